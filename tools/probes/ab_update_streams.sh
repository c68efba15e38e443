run() { python - "$@" <<'P'
import sys, json, io, contextlib
flag = sys.argv[1]
sys.argv = ["bench.py", "--steps", "10", "--warmup", "4", "--no-legs", "--no-cpu-baseline", "--no-kernels", "--no-host-probe"]
sys.path.insert(0, ".")
import egopose_amd.lstm as l, egopose_amd.agent as a
l.WGRAD_SIDE_STREAMS = flag[0] == "1"
a.AgentPPO.two_stream_heads = flag[1] == "1"
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1]); c = d["config"]
print(flag, round(d["value"]), c["t_sample_ms_median"], c["t_update_ms_median"])
P
}
for f in 11 01 00 11 01 00; do run $f 2>/dev/null | tail -1; done
