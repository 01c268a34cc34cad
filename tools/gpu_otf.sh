#!/bin/bash
timeout 600 python -m pytest tests/test_split_plane_gpu.py -m gpu -q -p no:cacheprovider -k "on_the_fly or volume_free" 2>&1 | tail -2
for i in 1 2; do python tools/bench_otf.py --pairs 79 --reps 5 2>&1 | grep OTF_; done
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/otf_fetch --output-format csv -- python $R/tools/bench_otf.py --pairs 79 --reps 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/otf_write --output-format csv -- python $R/tools/bench_otf.py --pairs 79 --reps 2 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
for name in ("otf_fetch", "otf_write"):
    tot = n = 0
    for f in glob.glob(f"gpurun_out/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "corr_otf_split" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    print(name, "dispatches", n, "counter per launch", tot / max(n, 1))
PY
rm -rf gpurun_out/otf_fetch gpurun_out/otf_write
