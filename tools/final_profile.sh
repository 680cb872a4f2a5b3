# tools/final_profile.sh [outdir]: the round's evidence from ONE build on ONE box -- rocprofv3 kernel stats of the bench command, the PMC passes of the same
# command (separate --pmc runs, kernel-trace only), their JSON (which bench.py reads for traffic / valu_frac_pmc / bound), then the bench line itself.
set -x
OUT=${1:-gpurun_out/r04/final}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu > $OUT/stats.log 2>&1 || exit 1
tools/pmc_pass.sh $OUT/pmc "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE TA_BUSY_avr GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" || exit 1
python3 tools/pmc_to_json.py $OUT/pmc $OUT/pmc.json || exit 1
python3 tools/kernel_stats_by_grid.py $OUT/stats > $OUT/kernel_stats_by_grid.csv
cp $OUT/pmc.json profiles/r04_pmc.json          # (the box's copy of the tree: the bench line below and the counters are then one build's, one box's)
timeout -k 10 700 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
tail -c 400 $OUT/bench_line.json
