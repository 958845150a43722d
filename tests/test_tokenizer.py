"""CPU: SPM tokenizer known answers of the reference's own test (tests/test-tokenizer-0-llama.cpp:12-40) on the reference's
vocab fixture (models/ggml-vocab-llama.gguf, committed under tests/golden/ as a data file), plus round trips."""
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
