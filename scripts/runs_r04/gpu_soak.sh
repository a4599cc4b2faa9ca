# randomised soaks on the final tree: AnyMDP step / token kernels against the oracle, the speculated filter against the direct one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYTHONPATH=.:tests timeout 1200 python tests/soak_anymdp.py ${1:-300} ${2:-4} > gpurun_out/r04_soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -3 gpurun_out/r04_soak_anymdp.txt | cut -c1-300; grep -c "^ok" gpurun_out/r04_soak_anymdp.txt
