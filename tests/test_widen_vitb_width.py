"""ViT-B/16 width (hidden 768, 12 heads: two passes of the row-dot wave, 12-head attention slices) through the ViT engine, forward and
every parameter gradient against HuggingFace's ViTModel - the same check as tests/test_emu_vit_engine.py at ViT-S-like width.  Kept in
its own late-sorting file: it was added after the last device run of round 1."""

from tests.test_emu_vit_engine import check_vit_engine_vs_hf


def test_vit_engine_forward_backward_vs_hf_vitb_width(stack_backend):
    check_vit_engine_vs_hf(stack_backend, 768, 1, 12, 768)
