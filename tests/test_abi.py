"""CPU: the C-ABI library builds for gfx950, loads, and exports every function include/fitclip_hip.h declares; the
ctypes table binds exactly that set.  No compute calls (no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    from fitclip_amd import build
    return build.build(verbose=False)


def _declared():
    text = (REPO / "include" / "fitclip_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound(lib_path):
    from fitclip_amd import _lib
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(str(lib_path))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in fitclip_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names


def test_error_reporting_without_gpu(lib_path):
    from fitclip_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.fc_version()
    cfg = _lib.fc_config(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12, 7, 0, 0, 0)  # precision 7 is invalid
    h = ctypes.c_void_p()
    assert lib.fc_create(cfg, h) == -1
    assert b"precision" in lib.fc_last_error()
    cfg.precision = _lib.PREC_BF16
    assert lib.fc_create(cfg, h) == 0
    try:
        assert lib.fc_num_weights(h) == 301
        assert lib.fc_weight_name(h, 0) == b"positional_embedding"
        shape = (ctypes.c_int64 * 1)(3)
        assert lib.fc_set_weight(h, b"nonsense.weight", 256, shape, 1) == -1
        assert b"unexpected key" in lib.fc_last_error()
        assert lib.fc_set_weight(h, b"ln_final.weight", 256, shape, 1) == -1
        assert b"shape mismatch" in lib.fc_last_error()
        assert lib.fc_set_weight(h, b"logit_scale", None, None, 0) == 0          # accepted and ignored
        assert lib.fc_pack_weights(h, 256, 1 << 40, None) == -4                   # weights missing
        assert lib.fc_encode_image(h, 256, 1, 256, 256, 1 << 30, None) == -4      # not packed
        # bf16 arena = every GEMM weight in bf16: 12 * (12 * 768^2) + 12 * (12 * 512^2) + conv + 2 projections
        want = 2 * (12 * 12 * 768 * 768 + 12 * 12 * 512 * 512 + 768 * 768 + 768 * 512 + 512 * 512)
        assert lib.fc_packed_bytes(h) == want
        assert lib.fc_workspace_bytes(h, 0, 1) > 197 * 768 * 4
    finally:
        lib.fc_destroy(h)


def test_parameter_names_match_reference_layout():
    import torch
    from fitclip_amd import synth
    from fitclip_amd.clip_model import CLIP, dims_from_state_dict
    from fitclip_amd.encoder import ClipVideoTextEncoder
    m = CLIP(synth.TINY)
    names = [n for n, _ in m.named_parameters()]
    assert names[:3] == ["positional_embedding", "text_projection", "logit_scale"]
    enc = ClipVideoTextEncoder(m, num_frames=4)
    names = [n for n, _ in enc.named_parameters()]
    assert names == ["model." + k for k in synth.parameter_shapes(synth.TINY)]
    assert "model.logit_scale" not in names and enc.model.visual.input_resolution == 64
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.TINY, 1).items()}
    assert dims_from_state_dict(sd) == synth.TINY
    enc.model.load_state_dict({**sd, "logit_scale": torch.tensor(1.0), "input_resolution": torch.tensor(64)})
    assert torch.equal(enc.model.ln_final.weight, sd["ln_final.weight"])


def test_product_fails_loudly_without_gpu():
    import torch
    from fitclip_amd import _lib, synth
    from fitclip_amd.clip_model import CLIP
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = CLIP(synth.TINY)
    with pytest.raises(_lib.FitclipHipError):
        m.encode_image(torch.zeros(1, 3, 64, 64))
    from fitclip_amd import ops
    with pytest.raises(_lib.FitclipHipError):
        ops.similarity(torch.zeros(4, 32), torch.zeros(4, 32))


def test_plugin_contract_table_matches_the_reference_members():
    """`aligner/encoder/video_encoder.py:14-52` + `video_text_encoder.py:15-31`: member names, forward dispatch and the
    NotImplementedError behaviour of members a subclass leaves open."""
    import torch

    from fitclip_amd.encoder import ClipVideoTextEncoder
    from fitclip_amd.plugin_api import TEXT_CONTRACT, VIDEO_CONTRACT, VideoEncoder, VideoTextEncoder

    assert set(VIDEO_CONTRACT) == {"encode_video", "get_train_frame_sampler", "get_eval_frame_sampler",
                                   "get_train_transform", "get_eval_transform", "to_bchw", "denormalize_video_tensor"}
    assert set(TEXT_CONTRACT) == {"encode_text", "get_tokenizer", "decode_text"}
    assert ClipVideoTextEncoder.missing_members() == []  # the shipped encoder implements the whole contract

    class VideoOnly(VideoEncoder):
        def encode_video(self, video):
            return video.flatten(1).sum(1, keepdim=True)

    class Half(VideoTextEncoder):
        def encode_video(self, video):
            return video.flatten(1).sum(1, keepdim=True)

    v = torch.ones(2, 1, 3, 2, 2)
    assert VideoOnly()(v).shape == (2, 1)
    assert "encode_video" not in VideoOnly.missing_members() and "encode_text" not in VideoOnly.missing_members()
    assert "encode_text" in Half.missing_members()
    with pytest.raises(NotImplementedError):
        Half()(video=v, text={"input_ids": torch.zeros(2, 4, dtype=torch.long)})
    with pytest.raises(NotImplementedError):
        Half().should_pad_batch
    with pytest.raises(NotImplementedError):
        VideoOnly().get_eval_frame_sampler()
