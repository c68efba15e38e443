import os, resource
print("cpus", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
try:
    os.nice(-10); print("nice ok ->", os.nice(0))
except Exception as e: print("nice fail", e)
for pol, name in ((os.SCHED_FIFO, "FIFO"), (os.SCHED_RR, "RR")):
    try:
        os.sched_setscheduler(0, pol, os.sched_param(1)); print(name, "ok"); os.sched_setscheduler(0, os.SCHED_OTHER, os.sched_param(0))
    except Exception as e: print(name, "fail", e)
print("rtprio rlimit", resource.getrlimit(resource.RLIMIT_RTPRIO), "nice rlimit", resource.getrlimit(resource.RLIMIT_NICE))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.weight", "/sys/fs/cgroup/cpu/cpu.rt_runtime_us", "/proc/sys/kernel/sched_rt_runtime_us", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "n/a", type(e).__name__)
