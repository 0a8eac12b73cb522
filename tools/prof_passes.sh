#!/bin/bash
# usage: tools/prof_passes.sh <tag> <cfg> <calls> [frames per call] -- runs rocprofv3 PMC passes, one run each (never combined with traces)
set -u
export TMPDIR=/tmp
tag=$1; cfg=$2; frames=$3; batch=${4:-1}
out=gpurun_out/$tag
mkdir -p $out
pass() { name=$1; shift; timeout -s KILL ${PASS_TIMEOUT:-90} rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_frame.py $cfg $frames $batch > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LEVEL_WAVES
pass sq2 SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64
pass sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
pass spi1 SPI_RA_TMP_STALL_CSN SPI_RA_VGPR_SIMD_FULL_CSN
pass spi2 SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
# HBM bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in passes of their own (FETCH_SIZE is doubled on gfx950 by prof_summary.py)
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_EA0_WRREQ_sum
for p in sq1 sq2 sqc spi1 spi2 tcc grbm fetch write; do python3 tools/prof_report.py $out/$p | grep -E "k_"; done
