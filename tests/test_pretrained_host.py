"""Host-side pieces of the real-weights path (no GPU): the CLIP BPE tokenizer on local vocabulary files, and the HuggingFace ->
OpenAI CLIP key mapping.  The vocabulary / weights are script-written stand-ins with the real structure (tests/pretrained_fixture.py):
the released files are not in the container - the algorithm is pinned by its published definition, ids of the real
vocabulary are parity-unpinned."""
import torch

from tests.pretrained_fixture import openai_to_hf, write_vocab
from xmask3d_amd import bpe as bpe_mod
from xmask3d_amd import checkpoint, clip_model


def test_bpe_merges_words_and_falls_back_to_bytes(tmp_path):
    write_vocab(str(tmp_path))
    tok = bpe_mod.ClipBPE.from_dir(str(tmp_path))
    assert tok.sot == 49406 and tok.eot == 49407 and len(tok.encoder) == 49408
    ids = tok.encode("The  CHAIR, seen&amp;nbsp;from frame 7!")
    toks = [tok.decoder[i] for i in ids]
    # known words are one token each (lower-cased, whitespace collapsed, html unescaped twice), punctuation / digits are byte symbols
    assert toks[:2] == ["the</w>", "chair</w>"] and ",</w>" in toks and "7</w>" in toks and "!</w>" in toks
    assert "seen</w>" in toks and "frame</w>" in toks
    # a word without merges: its letters, the last one word-final
    assert [tok.decoder[i] for i in tok.encode("zq")] == ["z", "q</w>"]
    # non-ASCII goes through the byte table (two UTF-8 bytes -> two symbols)
    assert len(tok.encode("é")) == 2
    assert tok.decode(tok.encode("the chair")).strip() == "the chair"


def test_tokenize_layout_padding_and_truncation(tmp_path):
    write_vocab(str(tmp_path))
    tok = bpe_mod.ClipBPE.from_dir(str(tmp_path))
    t = tok(["", "the wall"], context_length=77)
    assert t.shape == (2, 77) and t.dtype == torch.long
    assert t[0, :2].tolist() == [49406, 49407] and int(t[0, 2:].abs().sum()) == 0  # open_clip: zero padding
    assert t[1, 0] == 49406 and t[1, 3] == 49407 and t[1].argmax() == 3
    hf = tok([""], pad_id=tok.eot)
    assert hf[0, 0] == 49406 and bool((hf[0, 1:] == 49407).all())  # HuggingFace / ldm: padded with end-of-text
    long = tok(["wall " * 200], context_length=77)
    assert long.shape == (1, 77) and long[0, -1] == 49407 and long[0, 0] == 49406  # truncated, EOT kept last


def test_hf_clip_layout_maps_onto_the_openai_layout():
    torch.manual_seed(0)
    small = clip_model.CLIP(embed_dim=32, vision_layers=2, vision_width=64, vision_heads=4, text_layers=2, text_width=48, text_heads=4)
    sd = {k: v.clone() for k, v in small.state_dict().items()}
    hf = openai_to_hf(sd)
    assert "text_model.encoder.layers.1.self_attn.k_proj.weight" in hf and "vision_model.pre_layrnorm.weight" in hf
    back = checkpoint.map_hf_clip_state_dict(hf)
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    # text side only, under the prefix the Stable-Diffusion checkpoint uses
    text = {k: v for k, v in sd.items() if not k.startswith("visual.") and k not in ("logit_scale", "text_projection")}
    cond = checkpoint.map_hf_clip_state_dict(openai_to_hf(text, "cond_stage_model.transformer."), prefix="cond_stage_model.transformer.")
    assert set(cond) == set(text) and all(torch.equal(cond[k], text[k]) for k in text)
