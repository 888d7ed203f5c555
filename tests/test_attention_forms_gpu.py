"""The alternative forms of the fp32 attention kernels against each other -- a second net under the direct fp64 tests of
tests/test_kernels_gpu.py (test_attention_fwd_bwd / test_attention_dropout run every form, the one-wave dK/dV kernel included, against fp64).

The one-wave-per-SIMD kernels (attention.hip: attn_fwd_w1_body, attn_bwd_dq_w1_body, attn_bwd_dkv_w1_body) replace the two-wave / wave-pair
kernels for launches that meet their conditions (unpadded d = 128 / d = 64 heads, fp32 operands; dK/dV: at least two rounds of wave slots).
tools/attn_form_check.py runs seventeen shapes (sequences up to the 512 the one-wave kernels take; ragged tiles, fully masked rows, dropout on
and off, single tile, padded head dimensions, 576 keys) under two settings of the library's run-time options and compares tensor by tensor."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
def test_one_wave_forward_matches_the_two_wave_form():
    import attn_form_check as F
    # (the baseline's d-split of half-filled workgroups sums the head dimension in two halves: switched off for an order-for-order comparison)
    assert F.compare({"ATTN_W1": 0, "ATTN_DSPLIT": 0}, {"ATTN_W1": 1, "ATTN_DSPLIT": 0}) == 0


@pytest.mark.gpu
def test_one_wave_dq_matches_the_two_wave_form():
    import attn_form_check as F
    assert F.compare({"ATTN_W1": 1}, {"ATTN_W1": 3}, bwd=True) == 0


@pytest.mark.gpu
def test_one_wave_dkv_matches_the_wave_pair_form():
    import attn_form_check as F
    assert F.compare({"ATTN_W1": 1}, {"ATTN_W1": 5, "ATTN_W1_DKV_ANY": 1}, bwd=True) == 0


@pytest.mark.gpu
def test_all_one_wave_kernels_match_the_two_wave_forms():
    import attn_form_check as F
    assert F.compare({"ATTN_W1": 0, "ATTN_DSPLIT": 0}, {"ATTN_W1": 7, "ATTN_W1_DKV_ANY": 1, "ATTN_DSPLIT": 0}, bwd=True) == 0
