#!/bin/bash
# Regenerate EVERY profiles/r<NN>_* file that DESIGN.md quotes, from the tree as it stands (no stale evidence).
#   on the GPU box, in two gpurun calls (each well under an hour):
#       gpurun --timeout 3000 -- 'bash tools/final_profiles.sh part1'      tests, the bench lines, the traces
#       gpurun --timeout 3000 -- 'bash tools/final_profiles.sh part2'      PMC packs, data-parallel code paths, the 10 M-item configuration
#   then, in the build container:  python tools/collect_profiles.py r06    copies gpurun_out/final_* to profiles/r06_* (and README rows)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
S="bash tools/gpu_session.sh final"
# (timelines: a step in the MIDDLE of a sampler chunk — under the tracer the host is barely ahead of the device, and the step in which it
#  launches the next chunk of 16 batches shows that as a gap)
export TL_STEP=103
traces1() {
    $S trace
    cp gpurun_out/final_timeline.txt gpurun_out/final_timeline_one_step_unstalled.txt; cp gpurun_out/final_kernel_stats.txt gpurun_out/final_kernel_stats_default.txt; cp gpurun_out/final_kernel_stats.csv gpurun_out/final_kernel_stats_default.csv
    $S "trace:--stall_ms 150"
    cp gpurun_out/final_timeline.txt gpurun_out/final_timeline_one_step.txt; cp gpurun_out/final_kernel_stats.csv gpurun_out/final_kernel_stats_stalled.csv
}
traces2() {
    $S "trace:TCAR_FORCE_DP=1 --dp_mode sharded --stall_ms 150"
    cp gpurun_out/final_timeline.txt gpurun_out/final_timeline_sharded_1rank.txt; cp gpurun_out/final_kernel_stats.csv gpurun_out/final_kernel_stats_sharded_1rank.csv
    $S "trace:TCAR_FORCE_DP=1 TCAR_SIM_WORLD=8 --dp_mode sharded --stall_ms 150"
    cp gpurun_out/final_timeline.txt gpurun_out/final_timeline_sharded_simworld8.txt
}
case $1 in
  traces) traces1; traces2 ;;
  part1|part1_benches)
    [ "$1" == part1 ] && $S tests smoke
    $S bench:default "bench:default_20_5:--steps 20 --warmup 5" "bench:resident_feed:--resident_feed --no_cpu_baseline --no_e2e"
    $S "bench:adressa:--config adressa" "bench:mind:--config mind"
    $S "bench:globo_bf16x3:--scoring bf16x3 --no_cpu_baseline --no_e2e" "bench:globo_bf16:--scoring bf16 --no_cpu_baseline --no_e2e" "bench:globo_f32:--scoring f32 --no_cpu_baseline --no_e2e"
    traces1
    ;;
  part2)
    $S pmc:fwdce2:3 pmc:dx2:1 pmc:de2:1
    bash tools/pmc_gather.sh > gpurun_out/final_pmc_gather.log 2>&1; tail -3 gpurun_out/final_pmc_gather.log
    $S "bench:dp1rank_sharded:TCAR_FORCE_DP=1 --dp_mode sharded --no_cpu_baseline --no_e2e" "bench:dp1rank_replica:TCAR_FORCE_DP=1 --dp_mode replica --no_cpu_baseline --no_e2e"
    for w in 2 4 8; do $S "bench:sharded_simworld$w:TCAR_FORCE_DP=1 TCAR_SIM_WORLD=$w --dp_mode sharded --no_cpu_baseline --no_e2e"; done
    $S "bench:2ranks_one_gpu_gloo_sharded:--gpus 2 --same_device --backend gloo --dp_mode sharded --steps 30 --no_cpu_baseline --no_e2e"
    # RCCL on the one GPU: a process group of one rank, every collective of the exchanges issued (round 6) — directly on the step's stream
    # (rccl.py, the default) and through torch.distributed
    $S "bench:dp1rank_sharded_rccl_direct:TCAR_FORCE_COLLECTIVES=1 --dp_mode sharded --no_cpu_baseline --no_e2e" "bench:dp1rank_sharded_rccl_pg:TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 --dp_mode sharded --no_cpu_baseline --no_e2e"
    $S "bench:dp1rank_replica_rccl_direct:TCAR_FORCE_COLLECTIVES=1 --dp_mode replica --scoring bf16x3 --no_cpu_baseline --no_e2e"
    traces2
    $S "py:tools/thread_probe.py -1 40"
    cp gpurun_out/final_py.log gpurun_out/final_thread_probe.txt
    python tools/small_det_bench.py 2>&1 | grep "^T =" > gpurun_out/final_small_det_bench.txt; tail -3 gpurun_out/final_small_det_bench.txt
    python tools/step_times.py sampler 2>&1 | grep -v amdgpu.ids > gpurun_out/final_short_form_steps.txt; head -2 gpurun_out/final_short_form_steps.txt
    bash tools/trace_T.sh; cp gpurun_out/T_timelines.txt gpurun_out/final_timelines_T7_T4_T1.txt
    #  the long buckets: kernel statistics + one step's timeline of a T = 10 and a T = 40 loop
    for T in 10 40; do bash tools/trace_T40.sh $T > /dev/null; cp gpurun_out/T40_timeline.txt gpurun_out/final_timeline_T$T.txt; cp gpurun_out/T40_kernel_stats.txt gpurun_out/final_kernel_stats_T$T.txt; done
    $S "bench:stress10m:--config stress10m --steps 10 --warmup 2 --no_cpu_baseline --no_e2e"
    ;;
  sharded)      # only what the catalog-sharded code path touches (after a change confined to tcar_shard_*)
    python -m pytest tests -q -m gpu -k "sharded or rccl or dp" 2>&1 | tail -3 | tee gpurun_out/final_tests_sharded.txt
    $S "bench:dp1rank_sharded:TCAR_FORCE_DP=1 --dp_mode sharded --no_cpu_baseline --no_e2e"
    for w in 2 4 8; do $S "bench:sharded_simworld$w:TCAR_FORCE_DP=1 TCAR_SIM_WORLD=$w --dp_mode sharded --no_cpu_baseline --no_e2e"; done
    $S "bench:2ranks_one_gpu_gloo_sharded:--gpus 2 --same_device --backend gloo --dp_mode sharded --steps 30 --no_cpu_baseline --no_e2e"
    $S "bench:dp1rank_sharded_rccl_direct:TCAR_FORCE_COLLECTIVES=1 --dp_mode sharded --no_cpu_baseline --no_e2e" "bench:dp1rank_sharded_rccl_pg:TCAR_FORCE_COLLECTIVES=1 TCAR_RCCL_DIRECT=0 --dp_mode sharded --no_cpu_baseline --no_e2e"
    traces2
    ;;
  *) echo "usage: $0 part1|part1_benches|part2|sharded|traces" ;;
esac
