# Lab: consecutive stream-K ranges of the fused Mlp on one XCD (-DMF_XCD_RANGES):  tools/lab/build_variant.sh mf_xcd "-DMF_XCD_RANGES" tr_mlp_fused.hip; bash tools/lab/mlp_xcd_ab.sh
for rep in 1 2; do
echo product; python tools/mlp_lab.py 50432 35328 70001 2>&1 | grep "M="
echo xcd-ranges; TOKENREDUCTION_HIP_LIB=tools/lab/libtr_mf_xcd.so python tools/mlp_lab.py 50432 35328 70001 2>&1 | grep "M="
done
