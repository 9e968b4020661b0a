#!/bin/bash
# GPU visit: tools/exp_conv.py for the in-tree library and every build_ab/libkgan_*.so named in VARIANTS (default: all),
# then a table variant x case (ratio to the in-tree build).
set -u
mkdir -p gpurun_out
OUT=gpurun_out/exp_conv.log
: > $OUT
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
# a variant is "<lib>" or "<lib>:<ENV>=<value>" (an environment switch of that build)
for v in ${VARIANTS:-$(ls build_ab/libkgan_*.so 2>/dev/null)}; do
    lib=${v%%:*}; envs=""; tag=$(basename $lib .so)
    if [ "$lib" != "$v" ]; then envs=${v#*:}; tag="$tag:${envs#KG_CONV_}"; fi
    env KG_LIB=$lib KG_EXP_TAG=$tag $envs timeout 300 python tools/exp_conv.py >> $OUT 2>&1 || echo "FAILED $v" >> $OUT
done
python - <<'PY'
import re, collections
rows = collections.OrderedDict(); tags = []
for l in open("gpurun_out/exp_conv.log"):
    if not l.startswith("RES "):
        if "FAILED" in l or "Error" in l: print(l.rstrip())
        continue
    tag, n, name, t = re.match(r"RES (\S+) N=(\d+) \| (.*?) \| ([\d.]+) us", l).groups()
    if tag not in tags: tags.append(tag)
    rows.setdefault((n, name), {})[tag] = float(t)
print("%-44s" % "case" + "".join("%22s" % t[-20:] for t in tags))
for (n, name), d in rows.items():
    base = d.get(tags[0])
    print("%-44s" % (f"N={n} {name}") + "".join(("%14.1f (%.3f)" % (d[t], d[t] / base)) if t in d and base else "%22s" % "-" for t in tags))
PY
