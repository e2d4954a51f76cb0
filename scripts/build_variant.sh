#!/bin/bash
# scripts/build_variant.sh NAME [-D...]: liblhw.so with extra compile flags -> learninghumanoidwalking_amd/variants/liblhw_NAME.so
# (kernel experiments: run with LHW_LIB=<that file>).  HFLAGS: flags for lhw_humanoid.hip / lhw_humanoid_rollout.hip only, STEPFLAGS: the scheduling strategy of
# lhw_humanoid_rollout_step.hip (defaults: the product's, _lib.EXTRA_FLAGS), UFLAGS: extra flags of the other (update / cartpole / API) sources.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=learninghumanoidwalking_amd/variants; mkdir -p $D/obj_$NAME
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -w"
for f in learninghumanoidwalking_amd/csrc/*.hip; do
  X="${UFLAGS-}"; case "$(basename $f)" in
    lhw_humanoid_rollout_step.hip) X="-mllvm -disable-machine-licm -ffp-contract=on ${STEPFLAGS--mllvm -amdgpu-sched-strategy=iterative-ilp}";;
    lhw_humanoid*.hip) X="${HFLAGS--mllvm -disable-machine-licm -ffp-contract=on -mllvm -amdgpu-sched-strategy=iterative-maxocc}";; esac
  $C $X "$@" -c $f -o $D/obj_$NAME/$(basename $f).o &
done
wait
$C -shared -o $D/liblhw_$NAME.so $D/obj_$NAME/*.o
rm -rf $D/obj_$NAME
echo built $D/liblhw_$NAME.so
