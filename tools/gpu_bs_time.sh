#!/bin/bash
# round 5: tools/exp_conv.py (the discriminator's 13 kg_conv shapes at 64 / 192 samples) - direct form against the bf16-split form
mkdir -p gpurun_out
OUT=gpurun_out/bs_time.log
: > $OUT
KG_EXP_TAG=direct timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT
for t in ${BS_TILES:-auto 0 2}; do
  if [ $t = auto ]; then KG_CONV_BS=1 KG_EXP_TAG=bs-auto timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT
  else KG_CONV_BS=1 KG_CONV_BS_TILE=$t KG_EXP_TAG=bs-$t timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT; fi
done
python - <<'PY'
import re, collections
rows = collections.OrderedDict(); tags = []
for l in open("gpurun_out/bs_time.log"):
    m = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us \| (\S+) TF \| chk (\S+)", l)
    if not m: continue
    tag, n, name, t, tf, chk = m.groups()
    if tag not in tags: tags.append(tag)
    rows.setdefault((n, name), {})[tag] = (float(t), chk)
print("%-40s" % "case" + "".join("%24s" % t for t in tags))
for (n, name), d in rows.items():
    base = d.get(tags[0])
    print("%-40s" % (f"N={n} {name}") + "".join(("%10.1f (%.3f) %s" % (d[t][0], d[t][0] / base[0], d[t][1][:6])) if t in d else "%24s" % "-" for t in tags))
PY
