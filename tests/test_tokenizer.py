"""CPU: tokenizer known answers of the reference's own tests on the reference's vocab fixtures (committed under tests/golden/ as data files):
SPM - tests/test-tokenizer-0-llama.cpp:12-40 on models/ggml-vocab-llama.gguf; byte-level BPE - tests/test-tokenizer-0-falcon.cpp:12-41 on
models/ggml-vocab-falcon.gguf; plus round trips."""
import os

import pytest

from pipeinfer_amd import llama as LL

VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ggml-vocab-llama.gguf")

KNOWN = {
    "": [],
    " ": [259], "  ": [1678], "   ": [268], "\t": [29871, 12], "\n": [29871, 13], "\t\n": [29871, 12, 13],
    "Hello world": [15043, 3186], " Hello world": [29871, 15043, 3186], "Hello World": [15043, 2787], " Hello World": [29871, 15043, 2787],
    " Hello World!": [29871, 15043, 2787, 29991], "Hello, world!": [15043, 29892, 3186, 29991], " Hello, world!": [29871, 15043, 29892, 3186, 29991],
    " this is 🦙.cpp": [29871, 445, 338, 29871, 243, 162, 169, 156, 29889, 8223],
    "w048 7tuijk dsdfhu": [281, 29900, 29946, 29947, 29871, 29955, 9161, 13535, 18031, 2176, 6905],
    "нещо на Български": [1538, 4851, 665, 1386, 29713, 1305],
    "Hello": [15043], " Hello": [29871, 15043], "  Hello": [259, 15043], "   Hello": [1678, 15043], "    Hello": [268, 15043],
    "    Hello\n    Hello": [268, 15043, 13, 1678, 15043], " (": [29871, 313],
    "🚀 (normal) 😶‍🌫️ (multiple emojis concatenated) ✅ (only emoji that has its own token)":
        [29871, 243, 162, 157, 131, 313, 8945, 29897, 29871, 243, 162, 155, 185, 30722, 243, 162, 143, 174, 30598, 313, 20787, 953, 3848, 275, 16125, 630, 29897,
         29871, 31681, 313, 6194, 953, 29877, 2397, 393, 756, 967, 1914, 5993, 29897],
}


@pytest.fixture(scope="module")
def model():
    lib = LL.api()
    p = lib.llama_model_default_params()
    p.vocab_only = True
    h = lib.llama_load_model_from_file(VOCAB.encode(), p)
    assert h
    m = LL.Model.__new__(LL.Model)
    m.lib, m.h, m.n_vocab = lib, h, lib.llama_n_vocab(h)
    return m


def test_known_answers(model):
    assert model.n_vocab == 32000
    for text, want in KNOWN.items():
        assert model.tokenize(text, add_bos=False) == want, text
        assert model.tokenize(text, add_bos=True) == [1] + want, text


def test_detokenize_round_trip(model):
    for text in KNOWN:
        toks = model.tokenize(text, add_bos=False)
        out = b"".join(model.token_to_piece(t) for t in toks).decode("utf-8")
        assert out == " " + text if text else out == ""


def test_special_tokens_partition(model):
    # the space prefix is only added when the FIRST fragment is raw text (llama.cpp:6556-6558): "Hello" after <s> is token 10994, not 15043
    assert model.tokenize("<s>Hello</s>", add_bos=False, special=True) == [1, 10994, 2]
    assert model.tokenize("Hello</s>", add_bos=False, special=True) == [15043, 2]
    assert model.tokenize("<s>Hello", add_bos=False, special=False)[0] != 1


# ---- byte-level BPE (Falcon): tests/test-tokenizer-0-falcon.cpp:12-41 --------------------------------------------------------------------
VOCAB_FALCON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ggml-vocab-falcon.gguf")
KNOWN_FALCON = {
    "": [], " ": [204], "  ": [258], "   ": [466], "\t": [192], "\n": [193], "\t\n": [19125],
    "Hello world": [9856, 1079], " Hello world": [23090, 1079], "Hello World": [9856, 2889], " Hello World": [23090, 2889], " Hello World!": [23090, 2889, 12],
    "Hello, world!": [9856, 23, 1079, 12], " Hello, world!": [23090, 23, 1079, 12],
    " this is 🦙.cpp": [414, 304, 3346, 111, 231, 25, 29247],
    "w048 7tuijk dsdfhu": [98, 55866, 204, 34, 16682, 7149, 36190, 6869, 11481],
    "нещо на Български": [150, 133, 6207, 151, 215, 150, 134, 5052, 133, 6279, 5052, 223, 151, 216, 49679, 123, 53110, 47043, 7795],
    "កាន់តែពិសេសអាចខលចេញ": [38154, 206, 38154, 126, 38154, 225, 167, 237, 217, 38154, 221, 167, 237, 208, 38154, 228, 38154, 127, 38154, 237, 167, 237, 207, 38154, 237,
                            38154, 107, 38154, 126, 38154, 211, 38154, 207, 38154, 233, 38154, 211, 167, 237, 207, 38154, 215],
    "🚀 (normal) 😶‍🌫️ (multiple emojis concatenated) ✅ (only emoji that has its own token)":
        [2571, 232, 206, 204, 19, 11003, 20, 8196, 126, 283, 219, 48778, 116, 13392, 204, 19, 51831, 732, 63209, 1741, 7955, 522, 20, 22438, 211, 204, 19, 7927, 53360, 325, 504,
         701, 946, 10930, 20],
    "Hello": [9856], " Hello": [23090], "  Hello": [204, 23090], "   Hello": [258, 23090], "    Hello": [466, 23090], "    Hello\n    Hello": [466, 23090, 742, 23090],
    "\n =": [1212, 40], "' era": [18, 4932],
}


@pytest.fixture(scope="module")
def falcon():
    lib = LL.api()
    p = lib.llama_model_default_params()
    p.vocab_only = True
    h = lib.llama_load_model_from_file(VOCAB_FALCON.encode(), p)
    assert h
    m = LL.Model.__new__(LL.Model)
    m.lib, m.h, m.n_vocab = lib, h, lib.llama_n_vocab(h)
    return m


def test_bpe_known_answers(falcon):
    assert falcon.n_vocab == 65024
    for text, want in KNOWN_FALCON.items():
        assert falcon.tokenize(text, add_bos=False) == want, text


def test_bpe_detokenize_round_trip(falcon):
    for text in KNOWN_FALCON:
        toks = falcon.tokenize(text, add_bos=False)
        assert b"".join(falcon.token_to_piece(t) for t in toks).decode("utf-8") == text


def test_bpe_contractions_and_runs(falcon):
    # every split of the hand-written GPT-2 pre-tokenizer must round-trip, whatever it does with apostrophes, digits and runs of spaces
    for text in ("I'm sure they're here, we've seen it'll work; don't", "a1b22c333 4444  55555   x", "tabs\t\tand\n\nnewlines  \n ", "'" * 4 + "s't'"):
        toks = falcon.tokenize(text, add_bos=False)
        assert b"".join(falcon.token_to_piece(t) for t in toks).decode("utf-8") == text


# ---- the reference's tests/test-tokenizer-1-{llama,bpe}.cpp: every token and every codepoint survives detokenize . tokenize ----------------------
def _detok_spm(m, toks):
    """common.cpp:1129-1147 llama_detokenize_spm: the first non-BOS piece loses its leading space"""
    out = b""
    for i, t in enumerate(toks):
        piece = m.token_to_piece(t)
        if ((toks[0] == 1 and i == 1) or (toks[0] != 1 and i == 0)) and piece[:1] == b" ":
            piece = piece[1:]
        out += piece
    return out


def _detok_bpe(m, toks):
    return b"".join(m.token_to_piece(t) for t in toks)


def test_spm_every_token_and_codepoint_round_trips(model):
    for i in range(model.n_vocab):                                               # test-tokenizer-1-llama.cpp:63-72
        s = _detok_spm(model, [i])
        try:
            text = s.decode("utf-8")
        except UnicodeDecodeError:
            continue                                                             # byte tokens that are not text on their own
        assert _detok_spm(model, model.tokenize(text, add_bos=False)) == s, i
    step = int(os.environ.get("PI_TOKENIZER_CP_STEP", "1"))
    for cp in list(range(0, 0xffff, step)) + list(range(0x10000, 0x110000, 7 * step)):      # :74-95 (the astral planes sampled every 7th codepoint)
        if 0xd800 <= cp <= 0xdfff or cp == 9601 or cp == 0:
            continue
        text = chr(cp)
        assert _detok_spm(model, model.tokenize(text, add_bos=False)) == text.encode("utf-8"), hex(cp)


def test_bpe_every_token_and_codepoint_round_trips(falcon):
    for i in range(falcon.n_vocab):                                              # test-tokenizer-1-bpe.cpp:65-80
        s = _detok_bpe(falcon, [i])
        try:
            text = s.decode("utf-8")
        except UnicodeDecodeError:
            continue
        if "\x00" in text:
            continue                                                             # (the C API takes the text by length, the ctypes wrapper by bytes: fine, but keep NULs out)
        assert _detok_bpe(falcon, falcon.tokenize(text, add_bos=False)) == s, i
    skip = set(range(0x03, 0x06)) | {0x0b, 0x11, 0x19} | set(range(0x13, 0x18)) | set(range(0x1c, 0x1f))      # :83 "the GPT2 tokenizer doesn't want to interfere with some ASCII control characters"
    for cp in range(1, 0xffff):
        if cp in skip or 0xd800 <= cp <= 0xdfff:
            continue
        text = " " + chr(cp)
        assert _detok_bpe(falcon, falcon.tokenize(text, add_bos=False)) == text.encode("utf-8"), hex(cp)
    for cp in range(0x10000, 0x40000, 5):                                        # :96-97 "restrict to assigned unicode planes" (sampled)
        text = chr(cp)
        assert _detok_bpe(falcon, falcon.tokenize(text, add_bos=False)) == text.encode("utf-8"), hex(cp)


# ---- against the reference's own tokenizers (llama.cpp compiled into oracle/_ref/libref_kv.so, see oracle/ref_kv_harness.cpp) -----------------------------
FUZZ_POOL = [" ", "  ", "\n", "\t", "a", "b", "Hello", "world", "the", " the", "ing", "1", "22", "333", "4444", "'s", "'t", "'re", "'ll", "n't", ".", ",", "!", "?", "...", "-", "_", "(", ")",
             "é", "ü", "ß", "Ж", "привет", "日本", "語", "中文", "한국어", "🦙", "😶‍🌫️", "✅", "́", "‍", " ", "　", "\x01", "\x7f", "<s>", "</s>", "<unk>", "<|endoftext|>",
             ">>TITLE<<", "İ", "ǅ", "ﬁ", "０", "²", "½", "\r\n", "\x0b"]


@pytest.mark.parametrize("which", ["llama", "falcon"])
def test_random_text_tokenizes_like_the_reference_tokenizer(which, model, falcon):
    """differential test beyond the reference's known answers: random mixtures of words, digit runs, contractions, white space of every kind, accented / CJK /
    emoji / combining / control characters and special-token texts go through llama_tokenize of the reference's llama.cpp and through ours, with and without
    BOS, with and without special-token parsing; plus every token's piece"""
    import ctypes as C
    import numpy as np
    from tests import oracle_lib as O
    r = O.ref_kv()
    if r is None:
        pytest.skip("oracle/_ref/libref_kv.so is not built here (needs the reference sources)")
    mine = model if which == "llama" else falcon
    ref = r.reftok_load((VOCAB if which == "llama" else VOCAB_FALCON).encode())
    assert ref and r.reftok_n_vocab(ref) == mine.n_vocab

    def ref_tok(text, bos, special):
        b = text.encode("utf-8")
        out = np.zeros(len(b) + 16, np.int32)
        n = r.reftok_tokenize(ref, b, len(b), out.ctypes.data, out.size, int(bos), int(special))
        return None if n == -1000000 else [int(x) for x in out[:n]]          # None: the reference throws ("byte not found in vocab")

    rng = np.random.default_rng(3 if which == "llama" else 4)
    compared = 0
    for trial in range(700):
        n = int(rng.integers(1, 12))
        if trial % 7 == 0:
            text = "".join(chr(int(c)) for c in rng.integers(32, 0x3000, n))
        else:
            text = "".join(FUZZ_POOL[int(i)] for i in rng.integers(0, len(FUZZ_POOL), n))
        for bos in (False, True):
            for special in (False, True):
                want = ref_tok(text, bos, special)
                if want is not None:
                    assert mine.tokenize(text, add_bos=bos, special=special) == want, (text, bos, special)
                    compared += 1
    assert compared > 2000
    buf = C.create_string_buffer(256)
    for t in range(mine.n_vocab):
        n = r.reftok_piece(ref, t, buf, 256)
        assert mine.token_to_piece(t) == buf.raw[:n], t
    r.reftok_free(ref)
