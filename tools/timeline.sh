#!/bin/bash
# Run ON THE GPU BOX: kernel timeline (start order, duration, gap to the previous kernel's end) of the LAST
# <n> kernels of a command.   tools/timeline.sh <tag> <n> <command...>   -> gpurun_out/timeline_<tag>.txt
TAG=$1; N=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
D=/tmp/tl_$TAG
rm -rf $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -o k -- "$@" > $D.log 2>&1
F=$(find $D -name "*kernel_trace.csv" 2>/dev/null | head -1); if [ -z "$F" ]; then tail -20 $D.log; exit 1; fi
python3 - "$F" "$N" <<'PY' | tee $ROOT/gpurun_out/timeline_$TAG.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]); rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev_end)/1e3:7.1f}  grid {r.get("Grid_Size_X","?"):>8s} wg {r.get("Workgroup_Size_X","?"):>4s} lds {r.get("LDS_Block_Size","?"):>6s} vgpr {r.get("VGPR_Count","?"):>4s}  {r["Kernel_Name"][:70]}')
    busy += e - s; prev_end = max(prev_end, e)
print(f'span {(prev_end-t0)/1e3:.1f} us, sum of kernels {busy/1e3:.1f} us')
PY
