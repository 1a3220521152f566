for rep in 1 2; do
echo product; python tools/mlp_lab.py 50432 35328 70001 2>&1 | grep "M="
echo xcd-ranges; TOKENREDUCTION_HIP_LIB=tools/lab/libtr_mf_xcd.so python tools/mlp_lab.py 50432 35328 70001 2>&1 | grep "M="
done
