#!/bin/bash
# round 5: kg_conv's full-slice loop on the bf16 matrix cores with three-term operand splits (KG_CONV_SPLIT=1; weights split
# once per workgroup at staging time): the kg_conv kernel tests, then the 13 discriminator shapes under the automatic plan and
# forced tiles, fp32 MFMA against split
set -u
mkdir -p gpurun_out
OUT=gpurun_out/split.log
: > $OUT
if [ -z "${SPLIT_SKIP_TESTS:-}" ]; then
  KG_CONV_SPLIT=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv and not aggconv and default" 2>&1 | tail -5 >> $OUT
fi
export KG_EXP_N=${KG_EXP_N:-64,192}
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
KG_CONV_SPLIT=1 KG_EXP_TAG=split timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for pl in ${SPLIT_PLANS:-1,1 2,1 0,1}; do
  KG_CONV_SPLIT=1 KG_CONV_PLAN=$pl KG_EXP_TAG=split-$pl timeout 300 python tools/exp_conv.py >> $OUT 2>&1
done
python - <<'PY' | tee gpurun_out/split_table.log
import re, collections
rows = collections.OrderedDict(); tags = []
for l in open("gpurun_out/split.log"):
    m = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us", l)
    if not m: continue
    tag, n, name, t = m.groups()
    if tag not in tags: tags.append(tag)
    rows.setdefault((n, name), {})[tag] = float(t)
print("%-40s" % "case" + "".join("%16s" % t[-14:] for t in tags))
for (n, name), d in rows.items():
    base = d.get(tags[0])
    print("%-40s" % (f"N={n} {name}") + "".join(("%8.1f (%.2f)" % (d[t], d[t] / base)) if t in d and base else "%16s" % "-" for t in tags))
PY
grep -v "^RES" $OUT | grep -v amdgpu | tail -8
