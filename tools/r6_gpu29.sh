#!/bin/bash
# round-6 north-star counter record: the BertBiAttention kernels (both directions, one launch per kernel, cfg-2 shapes) under rocprofv3 --pmc,
# one pass per counter set, kernel trace only -> gpurun_out/pmc_round6_co_attn.json (layout of profiles/round4_co_attn_pmc.json)
export TMPDIR=/tmp; mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
CASES=co bash tools/pmc_run.sh round6_co_attn "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS;SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" python $GRAFT_REPO_ROOT/tools/attn_bench.py
