#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh <tag>`: distil gpurun_out/<tag> into the committed summaries under profiles/.
#   bash tools/publish_profiles.sh r05_f
# (run the round once more afterwards and publish again if gauss_traffic.json / descriptor_counters.json / kernel_stats_top.json
#  changed: the bench lines copy numbers from them)
set -e
TAG=${1:?tag}; T=gpurun_out/$TAG; P=profiles
python tools/pmc_traffic.py $T/pmc_fetch $T/pmc_write > $T/traffic.json
python tools/make_profile_json.py $T $TAG > /dev/null
cp $T/bench_driver_command.json $P/${TAG}_bench_driver_command.json
cp $T/bench_default.json $P/${TAG}_bench_default.json
cp $T/bench_batch16.json $P/${TAG}_bench_batch16.json
cp $T/bench_ctx1.json $P/${TAG}_bench_contexts1_under_rocprof.json
cp $T/bench_ctxd.json $P/${TAG}_bench_contexts6_under_rocprof.json
cp $T/host_path.json $P/${TAG}_host_path.json
cp $T/clock.csv $P/${TAG}_clock.csv
cp $T/counters.csv $P/${TAG}_counters.csv
cp $T/overlap_ctx1.txt $P/${TAG}_overlap_contexts1.txt
cp $T/overlap_default.txt $P/${TAG}_overlap_contexts6.txt
cp $T/stats_ctx1/*/*kernel_stats.csv $P/${TAG}_kernel_stats_contexts1.csv
cp $T/stats_default/*/*kernel_stats.csv $P/${TAG}_kernel_stats_contexts6.csv
cp $T/stats_default/*/*memory_copy_stats.csv $P/${TAG}_memory_copy_stats_contexts6.csv
cp $T/stats_cfg4/*/*kernel_stats.csv $P/${TAG}_kernel_stats_configs4.csv
cp $T/counters_cfg4.csv $P/${TAG}_counters_configs4.csv
cp $T/cfg4.json $P/${TAG}_configs4_under_rocprof.json
cp $T/stats_match/*/*kernel_stats.csv $P/${TAG}_kernel_stats_matcher.csv
cp $T/counters_match.csv $P/${TAG}_counters_matcher.csv
cp $T/match.txt $P/${TAG}_matcher_under_rocprof.txt
cp $T/provenance.json $P/${TAG}_provenance.json
grep -l consistency_error $P/${TAG}_bench_*.json && echo "^ lines with consistency_error: run the round again" || echo "published $TAG"
