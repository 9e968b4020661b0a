#!/bin/bash
# round 5: kg_conv on the bf16 matrix cores with 3-term operand splits (KG_CONV_SPLIT=1): kernel tests + the 13 shapes
set -u
mkdir -p gpurun_out
OUT=gpurun_out/split.log
: > $OUT
if [ -z "${SPLIT_SKIP_TESTS:-}" ]; then
  KG_CONV_SPLIT=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv and not aggconv and default" 2>&1 | tail -15 >> $OUT
fi
export KG_EXP_N=${KG_EXP_N:-64,192}
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
KG_CONV_SPLIT=1 KG_EXP_TAG=split timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for pl in ${SPLIT_PLANS:-}; do
  KG_CONV_SPLIT=1 KG_CONV_PLAN=$pl KG_EXP_TAG=split-$pl timeout 300 python tools/exp_conv.py >> $OUT 2>&1
  KG_CONV_PLAN=$pl KG_EXP_TAG=base-$pl timeout 300 python tools/exp_conv.py >> $OUT 2>&1
done
python - <<'PY' | tee gpurun_out/split_table.log
import re, collections
rows = collections.OrderedDict(); tags = []
chk = {}
for l in open("gpurun_out/split.log"):
    if not l.startswith("RES "):
        continue
    m = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us \| (\S+) TF \| chk (\S+)", l)
    if not m: 
        m2 = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us", l)
        tag, n, name, t = m2.groups(); c = None
    else:
        tag, n, name, t, tf, c = m.groups()
    if tag not in tags: tags.append(tag)
    rows.setdefault((n, name), {})[tag] = float(t)
    chk.setdefault((n, name), {})[tag] = c
print("%-40s" % "case" + "".join("%18s" % t[-16:] for t in tags))
for (n, name), d in rows.items():
    base = d.get(tags[0])
    print("%-40s" % (f"N={n} {name}") + "".join(("%10.1f (%.2f)" % (d[t], d[t] / base)) if t in d and base else "%18s" % "-" for t in tags))
for (n, name), d in chk.items():
    if n == "64": print("chk", name, d)
PY
grep -v "^RES" $OUT | tail -30
