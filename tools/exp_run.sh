SH="l4.conv2 d4,l4.conv3,l3.conv2 d2,l4.conv1,aspp d12"
python -m pytest tests/test_hip_fullsize.py -q -x -k "adjoint" 2>&1 | tail -2
for v in old base nomfma nostage; do
  L=""; O=""
  if [ $v = old ]; then O="--opt conv_w4=0"; elif [ $v != base ]; then L="WSDL_LIB=$PWD/weaklysuperviseddl_amd/csrc/exp/libwsdl_$v.so"; fi
  echo "### $v"
  env $L python tools/conv_shapes_bench.py --shapes "$SH" --only fwd,dgrad --reps 20 $O | grep -v "^shape"
done
