#!/usr/bin/env bash
# usage (on the GPU box, repo root):  bash tools/profile_config.sh TAG CONFIG CASES_PER_LAUNCH ALGORITHMIC_BYTES [extra bench.py args]
# One rocprofv3 --kernel-trace --stats run and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, three SQ groups: the
# counters of MI355X_MICROARCH.md "rocprofv3 PMC slots") of `python3 bench.py --config CONFIG`, raw output under
# gpurun_out/prof/TAG_CONFIG/, then the summaries the numbers are quoted from:
#   gpurun_out/profiles/TAG_CONFIG_kernel_stats.csv   this package's kernels only
#   gpurun_out/profiles/TAG_CONFIG_pmc_summary.json   per-kernel medians of every counter + derived ratios; FETCH_SIZE doubled
#   gpurun_out/profiles/traffic_CONFIG.json           HBM bytes per launch in the form bench.py reads
# (gpurun returns gpurun_out/ only: copy the three into profiles/ afterwards)
# --pmc is never combined with a tracing domain other than --kernel-trace (gpurun refuses that).
set -euo pipefail
TAG="$1"; CFG="$2"; CASES="$3"; ALG="$4"; shift 4
OUT="gpurun_out/prof/${TAG}_${CFG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON=(--config "$CFG" --no-cpu-baseline --no-parity "$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 bench.py "${COMMON[@]}" --steps 50 --warmup 5 > "$OUT/stats.log" 2>&1
echo "[profile_config] $CFG stats pass done"
pass() {  # pass NAME counters...
  local name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python3 bench.py "${COMMON[@]}" --steps 5 --warmup 1 > "$OUT/$name.log" 2>&1
  echo "[profile_config] $CFG pmc pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
python3 tools/pmc_summary.py "$TAG" "$CFG" "$OUT" "$CASES" "$ALG"
