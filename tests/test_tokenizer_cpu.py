"""Host WordPiece tokenizer (dawn_tokenizer_*) vs the HuggingFace `tokenizers` package.

The reference tokenises with the Rust crate tokenizers 0.13.3 driven by all-MiniLM-L6-v2's tokenizer.json
(src/embedding/embedding_service.rs:88,101-113).  Neither that crate's sources nor the real tokenizer.json are
available offline, so the C++ restatement is pinned against the Python binding of the same library
(tokenizers 0.22 here) configured with the same components — BertNormalizer(clean_text, handle_chinese_chars,
strip_accents=None, lowercase), BertPreTokenizer, WordPiece("##", "[UNK]", 100), "[CLS] $A [SEP]", truncation 128 —
over a synthetic vocabulary, on hand-written edge cases and seeded random Unicode.  Ids must be identical."""
import json
import os
import random
import unicodedata

import numpy as np
import pytest

tokenizers = pytest.importorskip("tokenizers")
from tokenizers import Tokenizer as HFTokenizer, models, normalizers, pre_tokenizers, processors  # noqa: E402

SPECIALS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]

CORPUS = [
    "Hello, World! This is DawnSearch: a distributed web search engine.",
    "The quick brown fox jumps over the lazy dog's back; 1234567890 times?",
    "café naïve Ångström résumé coöperate São Paulo Zürich",
    "İstanbul'da IŞIK ve ışık, straße STRASSE ǅ ǆ Σίσυφος ΟΔΥΣΣΕΥΣ",
    "中文字符 and 日本語のテキスト mixed 한국어 텍스트 too 𠀀𪜀",
    "tabs\tand\nnewlines\r\nand nbsp em　ideographic line",
    "zero​width‍joiners﻿bom soft­hyphen \x00nul �repl \x07bell \x7fdel",
    "emoji 😀👍🏽 family 👨‍👩‍👧 flags 🇳🇱 math ∑∫√ currency €£¥$ arrows →⇒",
    "punctuation … — – ‘quotes’ “double” «guillemets» ¿qué? ¡sí! (parens) [brackets] {braces} a_b a-b a/b a\\b a|b ~`^",
    "x" * 100 + " " + "y" * 101 + " " + "z" * 250,
    "literal [CLS] and [SEP] and [MASK] and [PAD] and [UNK] tokens, also [cls] lowercase and [ SEP ]",
    "combining ạ́ ạ́ ế ṩ ṩ ḍ̇ q̣̇ ̈́ ཱི ཱུ ཱྀ",
    "ﬁ ﬂ ﬀ ligatures ① ② ½ ¼ ² ³ ℃ ℉ Å K Ω fullwidth ＡＢＣ１２３！？",
    "",
    " ",
    "!!!",
    "a",
    "word " * 200,
    "supercalifragilisticexpialidocious antidisestablishmentarianism pneumonoultramicroscopicsilicovolcanoconiosis",
    "ǅemal Ǆ ǆ ǈ ß ẞ ŉ ǰ ΐ ΰ և ẖ ẗ ẘ ẙ ẚ",
    "math 𝐀𝐁𝐂 𝒜 𝔄 fraktur, hebrew שָׁלוֹם arabic السَّلَامُ devanagari नमस्ते thai สวัสดี",
]


def _hf_normalize_words(texts):
    nz = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
    pt = pre_tokenizers.BertPreTokenizer()
    words = []
    for t in texts:
        words += [w for w, _ in pt.pre_tokenize_str(nz.normalize_str(t))]
    return words


def _fuzz_texts(seed, n):
    rng = random.Random(seed)
    pools = [(0x20, 0x7E), (0xA0, 0x24F), (0x300, 0x36F), (0x370, 0x3FF), (0x400, 0x4FF), (0x590, 0x6FF), (0x900, 0x97F),
             (0xE00, 0xE7F), (0x1100, 0x11FF), (0x1E00, 0x1FFF), (0x2000, 0x206F), (0x20A0, 0x20CF), (0x2100, 0x214F),
             (0x2190, 0x21FF), (0x2460, 0x24FF), (0x3000, 0x303F), (0x3040, 0x30FF), (0x3400, 0x3410), (0x4E00, 0x4E80),
             (0xAC00, 0xAC80), (0xD7A0, 0xD7A3), (0xF900, 0xF920), (0xFB00, 0xFB4F), (0xFE00, 0xFE0F), (0xFF00, 0xFF60),
             (0x1D400, 0x1D430), (0x1F600, 0x1F640), (0x20000, 0x20010), (0x2B810, 0x2B830), (0x2B910, 0x2B930),
             (0x2CEA0, 0x2CEB5), (0x2F800, 0x2F810), (0xE000, 0xE010), (0x1, 0x1F), (0x7F, 0x9F), (0xFFF0, 0xFFFD)]
    out = []
    for _ in range(n):
        s = []
        for _ in range(rng.randint(1, 60)):
            r = rng.random()
            if r < 0.25:
                s.append(rng.choice(" \t\n  ,.;!?-'\"()"))
            elif r < 0.55:
                s.append(chr(rng.randint(0x61, 0x7A)) if rng.random() < 0.7 else chr(rng.randint(0x41, 0x5A)))
            else:
                lo, hi = rng.choice(pools)
                cp = rng.randint(lo, hi)
                ch = chr(cp)
                if 0xD800 <= cp <= 0xDFFF or unicodedata.category(ch) == "Cn":
                    continue  # unassigned in this Python's Unicode: the two libraries' tables may differ
                s.append(ch)
        out.append("".join(s))
    return out


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    import dawnsearch_amd as dawn
    d = tmp_path_factory.mktemp("tok")
    texts = CORPUS + _fuzz_texts(1, 300)
    words = _hf_normalize_words(texts)
    rng = random.Random(7)
    pieces = set()
    for w in words:
        if len(w) > 100:
            continue
        if rng.random() < 0.5:
            pieces.add(w)                      # whole word known
        for i, ch in enumerate(w):             # single characters (both positions) for ~80 % of characters
            if (ord(ch) * 2654435761) % 10 < 8:
                pieces.add(ch)
                pieces.add("##" + ch)
        for _ in range(2):                     # random inner pieces
            if len(w) >= 3:
                a = rng.randint(0, len(w) - 2)
                b = rng.randint(a + 1, len(w))
                pieces.add(("##" if a else "") + w[a:b])
    vocab_list = SPECIALS + sorted(pieces - set(SPECIALS))
    vocab = {t: i for i, t in enumerate(vocab_list)}
    hf = HFTokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    hf.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                               lowercase=True)
    hf.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    hf.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                      special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])])
    hf.add_special_tokens(SPECIALS)
    hf.enable_truncation(max_length=128)
    tj = str(d / "tokenizer.json")
    hf.save(tj)
    vt = str(d / "vocab.txt")
    with open(vt, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab_list) + "\n")
    return dawn, hf, tj, vt, texts


def test_matches_hf_tokenizers_on_corpus_and_fuzz(setup):
    dawn, hf, tj, vt, texts = setup
    for path in (tj, vt):
        tk = dawn.Tokenizer(path)
        assert tk.vocab_size() == hf.get_vocab_size()
        bad = []
        for t in texts + _fuzz_texts(2, 700):
            want = hf.encode(t.replace("\x00", "")).ids
            got = tk.encode(t).tolist()
            if got != want:
                bad.append((t, got, want))
        assert not bad, bad[:3]


def test_truncation_and_batch_packing(setup):
    dawn, hf, tj, vt, texts = setup
    tk = dawn.Tokenizer(tj)
    long = "word " * 200
    ids = tk.encode(long)
    assert len(ids) == 128 and ids[0] == 2 and ids[-1] == 3
    assert ids.tolist() == hf.encode(long).ids
    tk2 = dawn.Tokenizer(tj, max_length=0)
    assert len(tk2.encode(long)) == 202
    tk3 = dawn.Tokenizer(vt, max_length=16)
    hf.enable_truncation(max_length=16)
    try:
        for t in texts[:12]:
            assert tk3.encode(t).tolist() == hf.encode(t.replace("\x00", "")).ids
    finally:
        hf.enable_truncation(max_length=128)
    flat, offs = tk.encode_batch(texts[:9])
    assert offs[0] == 0 and len(offs) == 10
    for b, t in enumerate(texts[:9]):
        assert flat[offs[b]:offs[b + 1]].tolist() == hf.encode(t.replace("\x00", "")).ids


def test_large_batches_are_encoded_by_the_pool_with_the_same_ids(setup):
    """Batches of 32 texts and more are split over the tokenizer's worker threads (tokenizer.cpp: EncodePool): the same ids in the same
    order as one by one, whatever the batch size and however often the pool is reused; two callers at once (the second finds the
    pool busy and encodes on its own thread); a tokenizer destroyed with its workers idle."""
    import threading
    dawn, hf, tj, vt, texts = setup
    tk = dawn.Tokenizer(tj)
    pool = (texts + _fuzz_texts(3, 400)) * 2
    one_by_one = [tk.encode(t).tolist() for t in pool]
    for B in (31, 32, 33, 64, 257, len(pool)):
        for rep in range(2):
            flat, offs = tk.encode_batch(pool[:B])
            assert len(offs) == B + 1 and offs[0] == 0
            assert all(flat[offs[b]:offs[b + 1]].tolist() == one_by_one[b] for b in range(B)), B
    results = [None, None]

    def call(i):
        results[i] = tk.encode_batch(pool[:300])

    th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for flat, offs in results:
        assert all(flat[offs[b]:offs[b + 1]].tolist() == one_by_one[b] for b in range(300))
    tk.close()
    tk4 = dawn.Tokenizer(vt)  # (never used for a large batch: no workers to join)
    tk4.close()


def test_errors(setup, tmp_path):
    dawn = setup[0]
    with pytest.raises(dawn.DawnError):
        dawn.Tokenizer(str(tmp_path / "missing.txt"))
    p = tmp_path / "novocab.json"
    p.write_text(json.dumps({"model": {"type": "BPE"}}))
    with pytest.raises(dawn.DawnError):
        dawn.Tokenizer(str(p))
    q = tmp_path / "nospecials.txt"
    q.write_text("a\nb\n")
    with pytest.raises(dawn.DawnError):
        dawn.Tokenizer(str(q))
