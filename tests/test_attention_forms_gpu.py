"""The alternative forms of the d = 128 attention kernels against the forms the rest of the GPU suite already pins to the oracle.

The one-wave-per-SIMD kernels (attention.hip: attn_fwd_w1_body, attn_bwd_dq_w1_body, attn_bwd_dkv_w1_body) replace the two-wave / wave-pair
kernels for launches that meet their conditions (unpadded d = 128 heads, fp32 operands; dK/dV: at least two rounds of wave slots).  The small
shapes of test_kernels_gpu.py do not all meet them, and the choice is read once per process, so this test drives tools/attn_form_check.py:
each form in its own process, same seeded inputs, seventeen shapes (d = 128 and d = 64; sequences up to the 512 the one-wave kernels take; ragged tiles, fully masked rows, dropout on and off, single
tile, padded head dimensions, 576 keys), outputs compared tensor by tensor."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_form_check.py")], env={**os.environ, **env}, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "-> OK" in r.stdout


@pytest.mark.gpu
def test_one_wave_forward_matches_the_two_wave_form():
    # (the baseline's d-split of half-filled workgroups sums the head dimension in two halves: switched off for an order-for-order comparison)
    _run({"KNOB": "YTVLN_ATTN_W1", "A": "0", "B": "1", "YTVLN_ATTN_DSPLIT": "0"})


@pytest.mark.gpu
def test_one_wave_dq_matches_the_two_wave_form():
    _run({"KNOB": "YTVLN_ATTN_W1_DQ", "A": "0", "B": "1", "BWD": "1", "YTVLN_ATTN_W1_DKV": "0"})


@pytest.mark.gpu
def test_one_wave_dkv_matches_the_wave_pair_form():
    _run({"KNOB": "YTVLN_ATTN_W1_DKV", "A": "0", "B": "2", "BWD": "1", "YTVLN_ATTN_W1_DQ": "0"})


@pytest.mark.gpu
def test_one_wave_kernels_for_d64_match_the_two_wave_forms():
    # forward, dQ and dK/dV at once: the knob switches the d = 64 instantiations of all three (d = 128 stays one-wave on both sides)
    _run({"KNOB": "YTVLN_ATTN_W1_D64", "A": "0", "B": "1", "BWD": "1", "YTVLN_ATTN_DSPLIT": "0", "YTVLN_ATTN_W1_DKV": "2"})
