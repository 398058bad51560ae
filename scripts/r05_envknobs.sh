#!/bin/bash
# Round 5: do any of the HIP runtime's own knobs move the launch-to-launch floor of the captured decode step?
# (the small launches of a layer sit at ~4.5 us each, profiles/r05_decode_launch_durations.txt).  Same box, same binary,
# the headline stage of bench.py under each setting; the number compared is its tokens/s.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_envknobs; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 170 python3 bench.py --stage headline --gpus 1 --steps 6 --warmup 2 > $O/$tag.json 2> $O/$tag.err
  python3 - "$tag" $O/$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:34s} {d['value']:8.1f} tok/s   {d['ms_per_step']:.2f} ms/request")
except Exception as e:
    print(f"{sys.argv[1]:34s} FAILED {e}")
PY
}
run base_a X=1
run opt_flush_0 AMD_OPT_FLUSH=0
run packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run graph_batch_1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run graph_batch_1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run kernarg_copy_opt_0 DEBUG_HIP_KERNARG_COPY_OPT=0
run sys_scope_signal_0 ROC_SYSTEM_SCOPE_SIGNAL=0
run base_b X=1
