#!/bin/bash
# Which instantiations of the kernel families do the GPU suite and the fuzzers ever select?  (verdict r04 item 5)
#   bash tools/variant_census.sh      on the GPU box -> gpurun_out/variants/{selected.txt,summary.md}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/variants
mkdir -p $O
rm -f $O/selected.txt
export MIDYN_VARIANT_LOG=$O/selected.txt
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
timeout 600 python tools/fuzz_routes.py --cases 200 --seed 0 > $O/fuzz_routes.log 2>&1
timeout 600 python tools/fuzz_solver.py --cases 60 --seed 0 > $O/fuzz_solver.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
unset MIDYN_VARIANT_LOG
python - <<'PY'
import collections, os, re, subprocess, sys
sys.path.insert(0, "tests")
import codeobj
out = os.path.join("gpurun_out", "variants")
picked = sorted(set(l.strip() for l in open(os.path.join(out, "selected.txt")) if l.strip()))
names = subprocess.run(["c++filt"], input="\n".join(codeobj.library_kernels("qiskit_dynamics_amd/libmidyn.so")), capture_output=True, text=True).stdout.split("\n")
built = collections.defaultdict(set)
for n in names:
    m = re.match(r"void midyn::(\w+)<(.*)>\(", n)
    if m:
        built[m.group(1)].add(m.group(2).replace("true", "1").replace("false", "0"))
sel = collections.defaultdict(set)
for p in picked:
    m = re.match(r"(\w+)<(.*)>", p)
    sel[m.group(1)].add(m.group(2))
with open(os.path.join(out, "summary.md"), "w") as f:
    f.write("| kernel family | instantiations built | selected by the GPU suite + fuzzers + bench | never selected |\n|---|---|---|---|\n")
    for fam in sorted(sel):
        never = sorted(built.get(fam, set()) - sel[fam])
        f.write(f"| {fam} | {len(built.get(fam, ()))} | {len(sel[fam] & built.get(fam, sel[fam]))} | {'; '.join(never) if never else '-'} |\n")
print(open(os.path.join(out, "summary.md")).read())
PY
