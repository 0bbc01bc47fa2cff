"""Host-side placement of one rank of a multi-GPU run (one process per GPU): CPU affinity and the torch intra-op thread count.

Eight ranks on one node each run a Python enqueue thread (~60 000 C-ABI calls per step), an OT worker thread (exp-3/4/5) and torch's
intra-op pool; left alone, every rank's pool is sized for the whole machine (256 logical CPUs on the pool's hosts) and the kernel scheduler
migrates the enqueue threads across sockets.  ``pin_rank`` gives rank r a contiguous share of ONE NUMA node's CPUs (nodes are dealt to ranks in
order: ranks 0..n/2-1 on node 0, the rest on node 1 on a two-socket host, which is how the eight GPUs of an MI355X node hang off the sockets)
and caps the intra-op pool at that share.  Must run before the first HIP call of the process (bench.py, train.py call it right after
parsing their arguments).  FD_NO_AFFINITY=1 leaves the process alone; a single-rank run is never pinned."""
import glob
import os


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def numa_nodes():
    """[[cpus of node 0], [cpus of node 1], ...] restricted to the CPUs this process may use; one pseudo-node when sysfs has none."""
    allowed = set(os.sched_getaffinity(0))
    nodes = []
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda p: int(p.rsplit("node", 1)[1])):
        try:
            cpus = [c for c in _parse_cpulist(open(os.path.join(d, "cpulist")).read()) if c in allowed]
        except OSError:
            continue
        if cpus:
            nodes.append(cpus)
    return nodes or [sorted(allowed)]


def rank_cpus(local_rank, local_world, nodes=None):
    """The CPUs rank ``local_rank`` of ``local_world`` is pinned to (pure function of the node layout: testable without touching the process)."""
    nodes = numa_nodes() if nodes is None else nodes
    nn = len(nodes)
    node = min(local_rank * nn // local_world, nn - 1)
    peers = [r for r in range(local_world) if min(r * nn // local_world, nn - 1) == node]
    cpus = nodes[node]
    k, n = peers.index(local_rank), len(peers)
    per = max(len(cpus) // n, 1)
    share = cpus[k * per:(k + 1) * per] if k < n - 1 else cpus[k * per:]
    return share or cpus


def pin_rank(local_rank, local_world, max_threads=8):
    """Pins the process and sizes torch's intra-op pool; returns (cpus, threads) or None when nothing was done."""
    if local_world <= 1 or os.environ.get("FD_NO_AFFINITY") is not None or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    threads = max(1, min(len(cpus), max_threads))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    import torch
    torch.set_num_threads(threads)
    return cpus, threads
