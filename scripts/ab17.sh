# scripts/ab17.sh <lib>...: library builds on the large-polynomial sets (wave family on request), one round each
set -u
cd $GRAFT_REPO_ROOT
for spec in "PARAM_MESSAGE_1_CARRY_7 512" "PARAM_MESSAGE_4_CARRY_4 512" "PARAM_MESSAGE_3_CARRY_4 1024" "PARAM_MESSAGE_7_CARRY_0 1024" "PARAM_MESSAGE_3_CARRY_3 4096" "PARAM_MESSAGE_2_CARRY_4 4096" "PARAM_MESSAGE_2_CARRY_3 4096"; do
for lib in "$@"; do
  echo -n "$lib "; CONCRETE_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 170 python scripts/family_bench.py $spec 2 2 | sed 's/ variant.*kernel pbs/ pbs/' || exit 1
done; done
