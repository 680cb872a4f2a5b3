set -x
mkdir -p gpurun_out/r04/final
export TMPDIR=/tmp
timeout -k 10 700 python3 bench.py > gpurun_out/r04/final/bench_line.json 2> gpurun_out/r04/final/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/final/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/r04/final/stats.log 2>&1 || exit 1
tools/pmc_pass.sh gpurun_out/r04/final/pmc "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE TA_BUSY_avr GRBM_GUI_ACTIVE" "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
python3 tools/pmc_to_json.py gpurun_out/r04/final/pmc gpurun_out/r04/final/pmc.json
python3 tools/kernel_stats_by_grid.py gpurun_out/r04/final/stats > gpurun_out/r04/final/kernel_stats_by_grid.csv
tail -3 gpurun_out/r04/final/pmc/pass*.log | tail -20
