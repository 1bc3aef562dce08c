"""A synthetic, full-size CLIP merges file (48 894 merges -> vocabulary 49 408, SOT 49 406, EOT 49 407), generated from
a seed: the published `bpe_simple_vocab_16e6.txt.gz` is not available offline.  Words are random syllable strings; each
word contributes the merges of a random binary bracketing of its characters (operands are always existing symbols), so
real merge chains of depth 5-10 exist and the word list can be tokenized down to single ids.  Data, not code copied from
anywhere; the same function runs in the build container (fixture generation against the reference's SimpleTokenizer)
and on the test box (the .gz is written to a temp dir, never committed)."""
import gzip
import random
from typing import List, Tuple

N_MERGES = 49152 - 256 - 2


def byte_alphabet() -> Tuple[str, ...]:
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, spill = [], 0
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(256 + spill))
            spill += 1
    return tuple(table)


def synthetic_words(seed: int, count: int) -> List[str]:
    rng = random.Random(seed)
    onsets = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "st", "tr", "ch", "sh",
              "pl", "br", "gr", "é", "ü", "с", "к", "日", "本", "ß", "α"]
    nuclei = ["a", "e", "i", "o", "u", "ai", "ea", "ou", "oo", "ie", "ä", "ö", "а", "о", "語"]
    codas = ["", "", "n", "r", "s", "t", "l", "ng", "nd", "st", "ck", "m", "н"]
    words = set()
    while len(words) < count:
        syll = rng.choice((1, 1, 2, 2, 2, 3, 3, 4))
        words.add("".join(rng.choice(onsets) + rng.choice(nuclei) + rng.choice(codas) for _ in range(syll)))
    return sorted(words)


def write_synthetic_merges(path: str, seed: int = 0, n_merges: int = N_MERGES) -> List[str]:
    """Writes the gzip file and returns the word list it was built from."""
    rng = random.Random(seed)
    alphabet = byte_alphabet()
    words = synthetic_words(seed + 1, 40000)
    merges, seen = [], set()
    extra = ["'s", "'t", "'re", "ing", "ed", "123", "!!", "...", "<|", "|>"]
    for w in extra + words:
        sym = [alphabet[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>" if w not in ("<|",) else ""
        while len(sym) > 1 and len(merges) < n_merges:
            k = rng.randrange(len(sym) - 1)
            pair = (sym[k], sym[k + 1])
            if pair not in seen:
                seen.add(pair)
                merges.append(pair)
            sym[k:k + 2] = [sym[k] + sym[k + 1]]
        if len(merges) >= n_merges:
            break
    assert len(merges) == n_merges, len(merges)
    rng.shuffle(merges)  # ranks unrelated to the construction order: every application order gets exercised
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(('"bpe_simple_vocab_16e6.txt#version: synthetic"\n' + "\n".join(" ".join(m) for m in merges) + "\n").encode("utf-8"))
    return words
