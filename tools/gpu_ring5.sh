#!/bin/bash
# round 5: the ring form of kg_conv - parity (the kg_conv kernel tests with the ring forced, every ring tile) and timing
# at the 13 discriminator shapes for the direct kernel and every ring tile.   RING_TILES="0 1 2" RING_SKIP_TESTS=1
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ring5.log
: > $OUT
if [ -z "${RING_SKIP_TESTS:-}" ]; then
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv and not aggconv and (ring6 or ring7 or ring8 or ring9 or default)" 2>&1 | tail -15 >> $OUT
fi
export KG_EXP_N=${KG_EXP_N:-64,192}
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for t in ${RING_TILES:-6 7 8}; do
  KG_CONV_RING=1 KG_CONV_RING_TILE=$t KG_EXP_TAG=ring$t timeout 300 python tools/exp_conv.py >> $OUT 2>&1 || echo "FAILED ring$t" >> $OUT
done
python - <<'PY' | tee -a gpurun_out/ring5_table.log
import re, collections
rows = collections.OrderedDict(); tags = []
for l in open("gpurun_out/ring5.log"):
    if not l.startswith("RES "):
        continue
    tag, n, name, t = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us", l).groups()
    if tag not in tags: tags.append(tag)
    rows.setdefault((n, name), {})[tag] = float(t)
print("%-40s" % "case" + "".join("%16s" % t[-14:] for t in tags))
for (n, name), d in rows.items():
    base = d.get(tags[0])
    print("%-40s" % (f"N={n} {name}") + "".join(("%8.1f (%.2f)" % (d[t], d[t] / base)) if t in d and base else "%16s" % "-" for t in tags))
PY
grep -v "^RES" $OUT | tail -40
