# randomised soaks on the final tree: AnyMDP step / token kernels against the oracle, the speculated filter against the direct one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ "${1:-300}" = 0 ] || PYTHONPATH=.:tests timeout 1200 python tests/soak_anymdp.py ${1:-300} ${2:-4} > gpurun_out/r04_soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -3 gpurun_out/r04_soak_anymdp.txt | cut -c1-300; grep -c "^ok" gpurun_out/r04_soak_anymdp.txt
PYTHONPATH=.:tests timeout 1200 python tests/soak_linds.py ${3:-180} 5 > gpurun_out/r04_soak_linds.txt 2>&1; echo "soak linds rc=$?"; tail -3 gpurun_out/r04_soak_linds.txt | cut -c1-300; grep -c "^ok" gpurun_out/r04_soak_linds.txt
