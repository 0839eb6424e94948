#!/bin/bash
# Profiling passes of one round, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01_c'
# Writes under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards (see profiles/README.md).
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# what is being measured: the hash of the kernel sources of THIS snapshot (bench.py compares it with the sources it runs)
(cd $R && python3 -c "import json, time; from hessgpu_amd import build; print(json.dumps({'kernel_sources_sha16': build.sources_digest(), 'measured_utc': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}))" > $OUT/provenance.json)
# 0. the driver's own command, first thing on the fresh box (what BENCH_rNN.json will hold)
(cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err)
cd /tmp && export TMPDIR=/tmp
# 1. kernel statistics, single stream (averages comparable with bench.py's hipEvent roofline leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ctx1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > $OUT/bench_ctx1.json 2> $OUT/bench_ctx1.err
# 2. kernel statistics, default pipelined contexts (six)
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher > $OUT/bench_ctxd.json 2> $OUT/bench_ctxd.err
# 2b. the pipelined steps ALONE under the lightest trace (no statistics, no copy trace, no single-stream legs in the run):
#     what the overlap / idle summary is taken from (round 5: with the roofline leg's single-stream steps in the same trace
#     the "idle share" read 15 %; the pipelined part alone is 1 % idle)
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_pipelined -- python3 $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --no-profile --no-steady > /dev/null 2> $OUT/trace_pipelined.err
# 3./4. HBM traffic counters, one pass each
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > /dev/null 2> $OUT/pmc_write.err
# 5. wavefront occupancy of time (wait / issue-stall / active), one pass of 8 SQ counters
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > /dev/null 2> $OUT/pmc_sq.err
# 6. second SQ pass: LDS and memory-instruction counters
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > /dev/null 2> $OUT/pmc_sq2.err
# 7. effective shader clock per kernel (GRBM_GUI_ACTIVE / 8 / duration): keeps its kernel trace for the durations
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clk -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > /dev/null 2> $OUT/pmc_clk.err
# 8. configs[4] (4096^2, -topk 65536 -half, copier delivery: four descriptor launches): kernel statistics and the two SQ passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $R/tools/r06/cfg4_probe.py --quick --runs 10 --delivery dma > $OUT/cfg4.json 2> $OUT/cfg4.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_cfg4_sq -- python3 $R/tools/r06/cfg4_probe.py --quick --runs 2 --delivery dma > /dev/null 2> $OUT/pmc_cfg4_sq.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_cfg4_sq2 -- python3 $R/tools/r06/cfg4_probe.py --quick --runs 2 --delivery dma > /dev/null 2> $OUT/pmc_cfg4_sq2.err
python3 $R/tools/pmc_counters.py $OUT/pmc_cfg4_sq $OUT/pmc_cfg4_sq2 > $OUT/counters_cfg4.csv
# 9. the matcher: kernel statistics and counters at 8192^2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_match -- python3 $R/tools/bench_match.py > $OUT/match.txt 2> $OUT/match.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_match1 -- python3 $R/tools/bench_match.py --only 8192 > /dev/null 2> $OUT/pmc_match1.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_match2 -- python3 $R/tools/bench_match.py --only 8192 > /dev/null 2> $OUT/pmc_match2.err
python3 $R/tools/pmc_counters.py $OUT/pmc_match1 $OUT/pmc_match2 --kernels match > $OUT/counters_match.csv
python3 $R/tools/pmc_counters.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/counters.csv
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json 2> $OUT/traffic.err || true
python3 $R/tools/profile_clock_summary.py $OUT > $OUT/clock.csv 2> $OUT/clock.err || true
# concurrency summaries of the two kernel traces (before the traces are deleted)
python3 $R/tools/trace_overlap.py $OUT/stats_ctx1 > $OUT/overlap_ctx1.txt 2>&1 || true
python3 $R/tools/trace_overlap.py $OUT/trace_pipelined > $OUT/overlap_default.txt 2>&1 || true
# unprofiled bench lines on the same box: the default line and batches of 16
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --batch 16 --no-cpu-baseline --no-api-leg --no-configs4 --no-real-images --no-matcher > $OUT/bench_batch16.json 2> $OUT/bench_batch16.err
python3 tools/bench_host_path.py > $OUT/host_path.json 2> $OUT/host_path.err
# keep only the small summaries (the traces are large)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls -R $OUT | head -40
