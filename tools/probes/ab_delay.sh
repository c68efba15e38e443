run() { echo "== $*"; env "$@" timeout 200 python tools/probes/sample_time.py 9 2>&1 | grep "T_sample"; }
for i in 1 2; do
run EGP_CHAIN_DELAY_US=0
run EGP_CHAIN_DELAY_US=15
run EGP_CHAIN_DELAY_US=30
run EGP_CHAIN_DELAY_US=60
done
