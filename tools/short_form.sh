cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python bench.py --steps ${3:-20} --warmup ${4:-5} --no_cpu_baseline --no_e2e --no_kernel_timing $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; }
for r in 1 2 3; do
echo "r$r default 20/5:        $(run TCAR_X=0 '')"
echo "r$r resident 20/5:       $(run TCAR_X=0 --resident_feed)"
echo "r$r chunk4 20/5:         $(run TCAR_FEED_CHUNK=4 '')"
echo "r$r nodefer 20/5:        $(run TCAR_NO_DEFER=1 '')"
echo "r$r default 20/50:       $(run TCAR_X=0 '' 20 50)"
echo "r$r default 200/20:      $(run TCAR_X=0 '' 200 20)"
done
