"""ORACLE support (test infrastructure): how many host threads the CPU legs may use.

The GPU boxes expose 256 logical CPUs but the job may be confined by a cgroup quota; running
torch's CPU ops with one thread per *visible* CPU oversubscribes badly.  usable_cores() = the
smallest of (scheduler affinity, cgroup cpu.max quota, cap)."""
import math
import os


def usable_cores(cap: int = 32) -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.floor(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, cap))


def cpu_model() -> str:
    """CPU model string of the box (what `lscpu` prints as "Model name"), for the cpu_baseline record."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"
