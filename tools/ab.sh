#!/bin/bash
# ON THE GPU BOX: interleaved A/B of library builds.  One script for what round 4 did with 27 one-off ones.
#
#   tools/build_ab.sh quad "-DVCT_QUAD_SHARE=1" base ""          # (here) variants into build/ab/<name>.so
#   gpurun -- tools/ab.sh -t quad -l "tree quad base" -p "tests/test_gpu_parity.py tests/test_golden.py -k trace" \
#             -r 3 -f trace_kernel_ms,value -- "--scene atrium" "--scene bistro --voxel-dim 1024 --width 3840 --height 2160"
#
#   -l  libraries: `tree` = the in-tree libvct_amd.so, any other word = build/ab/<word>.so   (default: tree + all of build/ab)
#   -p  pytest selection run once per library before timing (parity first; omit to skip)
#   -r  rounds (default 2); libraries and argument sets are interleaved inside a round so drift hits all alike
#   -f  comma-separated keys of bench.py's JSON line to print (dotted paths allowed: extra.gi_pass_ms.total)
#   -s  steps per bench run (default 20)      -t  tag: results also go to gpurun_out/ab_<tag>.txt
#   --  one quoted bench.py argument set per word
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=""; PYTEST=""; ROUNDS=2; FIELDS="trace_kernel_ms,value"; STEPS=20; TAG=ab
while [ $# -gt 0 ]; do
  case "$1" in
    -l) LIBS="$2"; shift 2;; -p) PYTEST="$2"; shift 2;; -r) ROUNDS="$2"; shift 2;; -f) FIELDS="$2"; shift 2;;
    -s) STEPS="$2"; shift 2;; -t) TAG="$2"; shift 2;; --) shift; break;; *) echo "unknown option $1"; exit 2;;
  esac
done
[ -z "$LIBS" ] && LIBS="tree $(ls build/ab/*.so 2>/dev/null | xargs -n1 basename 2>/dev/null | sed 's/\.so$//')"
[ $# -eq 0 ] && set -- "--scene atrium"
mkdir -p gpurun_out
OUT=gpurun_out/ab_$TAG.txt; : > "$OUT"
use() { if [ "$1" = tree ]; then unset VCT_AMD_LIB; else export VCT_AMD_LIB="$PWD/build/ab/$1.so"; fi; }
if [ -n "$PYTEST" ]; then
  for lib in $LIBS; do use $lib
    echo "== $lib parity: $(timeout 1500 python -m pytest $PYTEST -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -1)" | tee -a "$OUT"
  done
fi
for round in $(seq 1 $ROUNDS); do for lib in $LIBS; do use $lib
  for args in "$@"; do
    timeout 600 python bench.py $args --steps $STEPS --warmup 5 --cpu-seconds 0 --no-sweep 2>/dev/null | grep "^{" | \
      python -c "
import sys, json
d = json.loads(sys.stdin.read())
def get(d, path):
    for k in path.split('.'):
        d = d.get(k) if isinstance(d, dict) else None
    return d
print('r$round', '$lib', '[$args]', ' '.join(f'{f}={get(d, f)}' for f in '$FIELDS'.split(',')))" | tee -a "$OUT"
  done
done; done
