run() { echo "== $*"; env "$@" timeout 200 python tools/probes/sample_time.py 9 2>&1 | grep "T_sample" | cut -c1-70; }
for r in 1 2; do
run EGP_PROBE_THREADS=14
run EGP_PROBE_THREADS=12
run EGP_PROBE_THREADS=15
run EGP_PROBE_THREADS=16
run EGP_PROBE_THREADS=20
done
