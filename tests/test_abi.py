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


def _header_struct_fields(name):
    text = (REPO / "include" / "fitclip_hip.h").read_text()
    body = re.search(r"typedef struct \{(.*?)\} %s;" % name, text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return re.findall(r"\b(int32_t|float|uint32_t|int64_t|size_t)\s+([a-zA-Z0-9_]+)\s*;", body)


def test_documented_binding_matches_the_header_struct():
    """INTEGRATION.md section 2 shows the reference-side ctypes stub.  Its FcConfig must list exactly the fields of the
    header's fc_config, in order, and its constructor call must pass one value per field with sizeof first; the package's
    own ctypes struct must agree too."""
    from fitclip_amd import _lib
    fields = _header_struct_fields("fc_config")
    assert fields[0] == ("int32_t", "struct_size") and all(t == "int32_t" for t, _ in fields)
    names = [n for _, n in fields]
    assert [n for n, _ in _lib.fc_config._fields_] == names
    assert ctypes.sizeof(_lib.fc_config) == 4 * len(names) and _lib.fc_config().struct_size == 4 * len(names)
    doc = (REPO / "INTEGRATION.md").read_text()
    stub = re.search(r"class FcConfig\(ctypes\.Structure\):.*?_fields_ = \[\(n, ctypes\.c_int32\) for n in \((.*?)\)\]", doc, flags=re.S)
    assert stub, "INTEGRATION.md no longer shows the FcConfig stub"
    documented = re.findall(r'"([a-z_0-9]+)"', re.sub(r"#.*", "", stub.group(1)))
    assert documented == names
    call = re.search(r"FcConfig\(ctypes\.sizeof\(FcConfig\),(.*?)\)\n", doc)
    assert call, "the documented constructor call must start with ctypes.sizeof(FcConfig)"
    assert len(call.group(1).split(",")) == len(names) - 1
    version = re.search(r"#define FC_ABI_VERSION (\d+)", (REPO / "include" / "fitclip_hip.h").read_text()).group(1)
    assert int(version) == _lib.ABI_VERSION and f"FC_ABI_VERSION {version}" in doc


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: the header must compile as C (no C++-isms), and FC_CONFIG_INIT must fill struct_size."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "fitclip_hip.h"\n'
                   "int main(void) {\n  fc_config c = FC_CONFIG_INIT;\n  fc_handle* h = 0;\n"
                   "  (void)h;\n  return c.struct_size == (int32_t)sizeof(fc_config) && c.embed_dim == 0 ? 0 : 1;\n}\n")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", str(REPO / "include"), str(src)],
                   check=True)
    exe = tmp_path / "use_header"
    subprocess.run([gcc, "-std=c99", "-I", str(REPO / "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_fc_create_rejects_a_binding_of_another_struct_revision(lib_path):
    """A caller compiled against an older fc_config (no struct_size; 15 or 16 int32 fields starting at embed_dim) or a future
    one is refused before any field is interpreted - never read past its end."""
    from fitclip_amd import _lib
    lib = _lib.load()
    assert lib.fc_version().endswith(b"abi %d" % _lib.ABI_VERSION)
    create = ctypes.CDLL(str(lib_path)).fc_create
    create.restype = ctypes.c_int32
    h = ctypes.c_void_p()
    values = [512, 224, 12, 768, 16, 77, 49408, 512, 8, 12, 0, 0, 0, 0, 0, 0]
    for n in (15, 16):  # the two pre-guard layouts: the first int32 the library sees is embed_dim = 512
        old = (ctypes.c_int32 * n)(*values[:n])
        assert create(ctypes.byref(old), ctypes.byref(h)) == -1 and not h.value
        msg = lib.fc_last_error()
        assert b"struct_size is 512" in msg and b"%d bytes" % ctypes.sizeof(_lib.fc_config) in msg, msg
    cfg = _lib.fc_config(*values)
    cfg.struct_size += 4                                      # a later revision with one more field
    assert lib.fc_create(cfg, h) == -1 and b"struct_size" in lib.fc_last_error()
    cfg.struct_size -= 4
    assert lib.fc_create(cfg, h) == 0
    lib.fc_destroy(h)


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
