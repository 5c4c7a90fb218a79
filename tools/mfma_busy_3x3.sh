#!/bin/bash
# MFMA utilisation of the 3x3 convolutions, one rocprofv3 --pmc run per (SURVEY.md 8a shape, pass):
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)      -> gpurun_out/mfma3x3/table.txt
# (north_star: ">= 40 % MFMA util on the 3x3 convs").  usage: tools/mfma_busy_3x3.sh [--opt name=value,...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma3x3
rm -rf $O; mkdir -p $O
# name:Cin,Cout,k,stride,dil,H,B (B = 16, 256x256 input)
SHAPES="l1.conv2:64,64,3,1,1,64,16 l2.0.conv2_s2:128,128,3,2,1,64,16 l2.conv2:128,128,3,1,1,32,16 l3.0.conv2/head3x3:256,256,3,1,1,32,16 l3.conv2_d2:256,256,3,1,2,32,16 l4.0.conv2_d2:512,512,3,1,2,32,16 l4.conv2_d4:512,512,3,1,4,32,16 aspp_d12:2048,256,3,1,12,32,16 aspp_d24:2048,256,3,1,24,32,16 aspp_d36:2048,256,3,1,36,32,16 aux3x3:1024,256,3,1,1,32,16"
for sh in $SHAPES; do
  name=${sh%%:*}; geo=${sh#*:}; tag=$(echo $name | tr '/.' '__')
  for pass in fwd dgrad wgrad; do
    if [ "$name" = "aux3x3" ] && [ "$pass" != "fwd" ]; then continue; fi
    rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${tag}_${pass} -- python3 $R/tools/one_conv.py --shape $geo --pass $pass --reps 5 "$@" > /dev/null 2>&1 || { echo "FAILED $name $pass"; exit 1; }
    echo "$name $pass $geo" >> $O/index.txt
    echo "[mfma3x3] $name $pass done" >&2
  done
done
# ASPP as the step runs it since round 5: the four branch convolutions as ONE grouped forward launch and ONE multi-source
# input-gradient launch (the single-branch rows above stay in the table for comparison and are left out of the fwd / dgrad totals)
for pass in fwd dgrad; do
  rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/aspp_4branches_${pass} -- python3 $R/tools/one_conv.py --shape aspp,16,32 --pass $pass --reps 5 "$@" > /dev/null 2>&1 || { echo "FAILED aspp_4branches $pass"; exit 1; }
  echo "aspp_4branches $pass aspp,16,32" >> $O/index.txt
  echo "[mfma3x3] aspp_4branches $pass done" >&2
done
python3 $R/tools/mfma_busy_table.py $O > $O/table.txt
cat $O/table.txt
