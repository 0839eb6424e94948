#!/bin/bash
# kernel statistics of the matcher bench:  tools/r06/match_prof.sh TAG
TAG=${1:-match}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/bench_match.py > $OUT/bench.txt 2> $OUT/bench.err
cat $OUT/bench.txt
python3 - $OUT <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    # per (kernel, grid) averages: the three sizes of the bench separate by grid size
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        agg[(n, r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in sorted(agg.items()):
        print(f"{k[0][:36]:36s} grid {k[1]:>8s} x {k[2]:>4s} calls {len(v):3d} avg_us {sum(v)/len(v):8.1f} min_us {min(v):8.1f}")
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
