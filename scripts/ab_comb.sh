# A/B of the bootstrap kernels that read an input combination against builds without that: 1_7 and 1_1
# take the block kernel under AUTO
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for lib in build/libconcrete_amd_oldblock.so concrete_amd/_lib/libconcrete_amd.so; do
  echo "== $lib"
  QB_PARAMS=PARAM_MESSAGE_1_CARRY_7 CONCRETE_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 170 python scripts/quick_bench.py 0 256 2 | head -1 | sed 's/.*kernel pbs/1_7 pbs/' || exit 1
  QB_PARAMS=PARAM_MESSAGE_1_CARRY_1 CONCRETE_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 170 python scripts/quick_bench.py 0 16384 3 | head -1 | sed 's/.*kernel pbs/1_1 pbs/' || exit 1
  QB_PARAMS=PARAM_MESSAGE_1_CARRY_2 CONCRETE_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 170 python scripts/quick_bench.py 0 32768 3 | head -1 | sed 's/.*kernel pbs/1_2 pbs/' || exit 1
done
done
