# scripts/ab_block.sh <lib>...: library builds on sets served by the block kernel (variant 3), one round each
set -u
cd $GRAFT_REPO_ROOT
for spec in "PARAM_MESSAGE_1_CARRY_7 256" "PARAM_MESSAGE_4_CARRY_4 256" "PARAM_MESSAGE_1_CARRY_1 16384" "PARAM_MESSAGE_2_CARRY_2 4096" "PARAM_MESSAGE_3_CARRY_3 1024"; do
for lib in "$@"; do
  echo -n "$lib "; CONCRETE_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 170 python scripts/family_bench.py $spec 2 3 | sed 's/ variant.*kernel pbs/ pbs/' || exit 1
done; done
