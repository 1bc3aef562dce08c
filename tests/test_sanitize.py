"""CPU sanitizer target (SURVEY.md section 5 "race detection / sanitizers"; never the GPU build): the host-only code of the library
- csrc/bpe.cpp, which parses an external gzip file and arbitrary UTF-8 - built with g++ -fsanitize=address,undefined
(`python -m fitclip_amd.build --host-asan`) and driven through its C ABI by tests/host/bpe_sanitize_driver.cpp.  Any sanitizer
report aborts the driver (non-zero exit)."""
import gzip
import json
import subprocess

import pytest

from fitclip_amd import build
from fitclip_amd.bpe import clean_text


@pytest.fixture(scope="module")
def driver():
    return str(build.build_host_asan(verbose=False))


def _run(args, **kw):
    res = subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-3000:])
    return res.stdout


@pytest.mark.parametrize("name", ["bpe_toy", "bpe_full"])
def test_fixture_ids_under_the_sanitizers(driver, golden_dir, tmp_path, name):
    """The reference-generated ids (slip.SimpleTokenizer; tests/golden/make_goldens.py) come out of the instrumented build."""
    g = json.loads((golden_dir / f"{name}.json").read_text())
    if name == "bpe_full":   # the full-size synthetic merges file is regenerated from its seed (never committed)
        import sys
        sys.path.insert(0, str(golden_dir))
        from synth_merges import write_synthetic_merges
        merges = str(tmp_path / "full.txt.gz")
        write_synthetic_merges(merges, seed=g["merges_seed"])
        sot, eot = g["sot"], g["eot"]
    else:
        merges = str(golden_dir / "bpe_toy_merges.txt.gz")
        sot, eot = g["vocab_size"] - 2, g["vocab_size"] - 1
    texts = [clean_text(t) for t in g["texts"]]
    assert not any("\n" in t for t in texts)
    (tmp_path / "texts.txt").write_text("\n".join(texts) + "\n", encoding="utf-8")
    out = _run([driver, merges, "texts", str(tmp_path / "texts.txt")])
    rows = [[int(v) for v in line.split()[1:]] for line in out.splitlines() if line.startswith("ids:")]
    assert len(rows) == len(texts)
    for row, ids in zip(rows, g["ids"]):   # the fixture holds SimpleTokenizer.encode's ids: framed, cut (EOT restored), padded
        want = [sot] + ids + [eot]
        if len(want) > 77:
            want = want[:76] + [eot]
        assert row == want + [0] * (77 - len(want))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hostile_inputs_pass_without_a_report(driver, golden_dir, seed):
    """Random bytes, truncated / overlong / surrogate UTF-8, special-token fragments, 100 KB single tokens, short output buffers,
    ids out of range in both directions - through encode, tokenize (with and without truncation) and decode."""
    out = _run([driver, str(golden_dir / "bpe_toy_merges.txt.gz"), "fuzz", str(seed), "400"])
    assert out.startswith("fuzz ok ")


def test_malformed_merge_files_are_refused_not_parsed(driver, golden_dir, tmp_path):
    good = gzip.decompress((golden_dir / "bpe_toy_merges.txt.gz").read_bytes())
    files = {}
    files["plain.txt"] = good                                                  # no gzip magic: zlib would read it as text
    files["garbage.bin"] = bytes(range(256)) * 64
    files["empty.gz"] = b""
    files["truncated.gz"] = (golden_dir / "bpe_toy_merges.txt.gz").read_bytes()[:200]
    files["magic_only.gz"] = b"\x1f\x8b"
    files["bad_alphabet.gz"] = gzip.compress(b"#version\n\xe4\xb8\xad \xe6\x96\x87\n" + good.split(b"\n", 1)[1])
    files["no_newline.gz"] = gzip.compress(b"#version: 0.2")
    files["binary_lines.gz"] = gzip.compress(b"#v\n" + bytes([0, 1, 2, 255, 254, 10, 32, 32, 10, 0xC3, 10]) * 50)
    paths = []
    for fname, data in files.items():
        (tmp_path / fname).write_bytes(data)
        paths.append(str(tmp_path / fname))
    paths.append(str(tmp_path / "does_not_exist.gz"))
    out = _run([driver, "-", "create", *paths, str(golden_dir / "bpe_toy_merges.txt.gz")])
    rcs = dict(zip([*files, "missing", "good"], [int(line.split()[1]) for line in out.splitlines()]))
    assert rcs["good"] == 0
    for bad in ("plain.txt", "garbage.bin", "empty.gz", "truncated.gz", "magic_only.gz", "bad_alphabet.gz", "missing"):
        assert rcs[bad] != 0, (bad, rcs)
    assert rcs["no_newline.gz"] == 0      # a header line alone: the 514-entry base vocabulary, as the reference would build


def test_product_library_refuses_a_file_without_the_gzip_magic(golden_dir, tmp_path):
    """The same check in the shipped library (not only in the instrumented build)."""
    from fitclip_amd import _lib
    from fitclip_amd.bpe import ClipBpeTokenizer
    plain = tmp_path / "merges.txt"
    plain.write_bytes(gzip.decompress((golden_dir / "bpe_toy_merges.txt.gz").read_bytes()))
    with pytest.raises(_lib.FitclipHipError, match="gzip"):
        ClipBpeTokenizer(str(plain))
    assert ClipBpeTokenizer(str(golden_dir / "bpe_toy_merges.txt.gz")).vocab_size > 514
