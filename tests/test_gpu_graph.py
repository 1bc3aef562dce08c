"""hipGraph capture of the encode-and-score path (include/fitclip_hip.h: "all functions may be captured into a hipGraph").

The reference-shaped call - one eval batch of 32 clips x 4 frames + 32 captions (aligner/data/video_data_module.py:32,
aligner/encoder/clip_video_text_encoder.py:69) - is captured ONCE on a side stream: `fc_encode_image` (visual tower on the
capture stream), `fc_encode_text` (text tower on the encoder's second stream: a fork / join inside the capture),
`fc_pool_normalize`, `fc_l2_normalize`, `fc_similarity`, `fc_ranks`, `fc_similarity_ranks`; the graph is replayed and every
output compared BITWISE with the eager run.  Capture goes through torch.cuda.CUDAGraph (hipStreamBeginCapture /
hipGraphInstantiate / hipGraphLaunch); the library sees nothing but the stream it is handed.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fitclip_amd import ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402

DEV = "cuda"


def _capture_and_replay(enc, video, ids, replays=3):
    text = {"input_ids": ids}

    def call():
        ev, et = enc(video=video, text=text)
        scores = ops.similarity(et, ev)
        return ev, et, scores, ops.ranks(scores), ops.similarity_ranks(et, ev)

    side = torch.cuda.Stream(device=video.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.inference_mode():
        eager = [t.clone() for t in call()]       # (first calls: weight packing, workspaces, dynamic-LDS attributes, side stream)
        eager2 = call()
        assert all(torch.equal(a, b) for a, b in zip(eager, eager2))
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            captured = call()
        for _ in range(replays):
            for t in captured:
                t.zero_()                          # a replay must rewrite every output
            graph.replay()
            side.synchronize()
            for name, a, b in zip(("video", "text", "scores", "ranks", "fused ranks"), eager, captured):
                assert torch.equal(a, b), name
    torch.cuda.current_stream().wait_stream(side)
    return eager


@pytest.mark.parametrize("precision", ["fp32", "fp32x6", "fp32x3", "bf16"])
def test_reference_shaped_call_replays_bitwise_from_a_hipgraph(vitb16_state_dict, precision):
    d = synth.VIT_B_16
    enc = ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision=precision, device=DEV), num_frames=4)
    video = torch.from_numpy(synth.make_video(32, 4, d, seed=3)).to(DEV)
    ids = torch.from_numpy(synth.make_text(32, d, seed=3)).to(DEV)
    ev, et, scores, ranks, fused = _capture_and_replay(enc, video, ids)
    assert ev.shape == (32, d.embed_dim) and et.shape == (32, d.embed_dim) and scores.shape == (32, 32)
    assert torch.equal(ranks, fused)
    assert torch.isfinite(ev).all() and torch.isfinite(et).all()


def test_plane_text_call_replays_bitwise_from_a_hipgraph(vitb16_state_dict):
    """fp32x3 with 64 captions per call (4928 token rows): the text tower's block GEMMs run on the three-product kernel, and the output
    scan + the range flag's copy to its pinned mirror at the end of `fc_encode_text` are captured with them."""
    d = synth.VIT_B_16
    enc = ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision="fp32x3", device=DEV), num_frames=2)
    video = torch.from_numpy(synth.make_video(64, 2, d, seed=4)).to(DEV)
    ids = torch.from_numpy(synth.make_text(64, d, seed=4)).to(DEV)
    ev, et, scores, ranks, fused = _capture_and_replay(enc, video, ids)
    enc.model.check_range()
    plain = ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision="fp32", device=DEV), num_frames=2)
    want = plain.encode_text({"input_ids": ids})
    assert not torch.equal(et, want) and float((et - want).abs().max()) < 2e-6       # the plane arithmetic, at fp32 accuracy
    assert torch.equal(ranks, fused) and torch.isfinite(ev).all()


def test_graph_with_new_inputs_in_the_captured_buffers(tiny_state_dict):
    """A captured graph reads its inputs from the addresses it was captured with: new data copied INTO those tensors must come
    out as the eager result for that data (the usual serving pattern: static input buffers, one graph launch per batch)."""
    d = synth.TINY
    enc = ClipVideoTextEncoder(build_clip(tiny_state_dict, precision="fp32", device=DEV), num_frames=2)
    video = torch.from_numpy(synth.make_video(6, 2, d, seed=1)).to(DEV)
    ids = torch.from_numpy(synth.make_text(6, d, seed=1)).to(DEV)
    text = {"input_ids": ids}
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.inference_mode():
        enc(video=video, text=text)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ev, et = enc(video=video, text=text)
        for seed in (2, 3):
            v2 = torch.from_numpy(synth.make_video(6, 2, d, seed=seed)).to(DEV)
            i2 = torch.from_numpy(synth.make_text(6, d, seed=seed)).to(DEV)
            video.copy_(v2)
            ids.copy_(i2)
            graph.replay()
            side.synchronize()
            want_v, want_t = enc(video=v2, text={"input_ids": i2})
            assert torch.equal(ev, want_v) and torch.equal(et, want_t)
    torch.cuda.current_stream().wait_stream(side)
