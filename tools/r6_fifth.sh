#!/bin/bash
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
python tools/stem_wgrad_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/stem_wgrad2.txt
for o in "tile_img_major=1" "tile_img_major=0" "tile_img_major=1,group_tps10=45" "tile_img_major=0,group_tps10=45" "tile_img_major=1,group_tps10=60"; do
  echo "### $o"; python tools/aspp_group_bench.py --opt $o 2>&1 | grep -E "^forward|^input"
done | tee gpurun_out/r6/aspp_sweep.txt
cd /tmp && export TMPDIR=/tmp
for o in 1 0; do for pass in fwd dgrad; do
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/r6/pmc_aspp_${pass}_im$o -- python3 $R/tools/one_conv.py --shape aspp,16,32 --pass $pass --reps 3 --opt tile_img_major=$o > /dev/null 2>&1
  f=$(ls $R/gpurun_out/r6/pmc_aspp_${pass}_im$o/*/*counter_collection.csv | head -1)
  python3 - "$f" "$pass img_major=$o" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in acc.items():
    if k.startswith("conv_igemm_split"):
        print(f"{sys.argv[2]:22s} {k[:60]:60s} launches {n}  FETCH_SIZE x2 = {2 * v / n * 1024 / 1e9:.2f} GB per launch")
PY
done; done | tee $R/gpurun_out/r6/aspp_traffic.txt
cd $R
bash tools/pmc_one.sh r6_l4conv2_fwd --shape 512,512,3,1,4,32,16 --pass fwd --reps 3 > gpurun_out/r6/pmc_l4conv2_fwd.txt 2>&1; grep -i "lds\|kernel" gpurun_out/r6/pmc_l4conv2_fwd.txt | cut -c1-200 | head -20
