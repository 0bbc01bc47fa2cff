"""Host-side placement of one rank of a multi-GPU run (one process per GPU): CPU affinity and the torch intra-op thread count.

Eight ranks on one node each run a Python enqueue thread (~60 000 C-ABI calls per step), an OT worker thread (exp-3/4/5) and torch's
intra-op pool; left alone, every rank's pool is sized for the whole machine (256 logical CPUs on the pool's hosts) and the kernel scheduler
migrates the enqueue threads across sockets.  ``pin_rank`` gives rank r a share of the CPUs of the NUMA node ITS GPU hangs off and caps the
intra-op pool at that share:

  * the node comes from sysfs -- KFD topology node k-th with SIMDs = HIP device k (``drm_render_minor`` ->
    ``/sys/class/drm/renderD<minor>/device/numa_node``; ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` index lists are honoured) -- and only
    when that cannot be read from the even deal of round 4 (ranks 0..n/2-1 on node 0, the rest on node 1 on a two-socket host);
  * the ranks of one node are dealt whole PHYSICAL cores (``topology/thread_siblings_list``): a cpulist like ``0-63,128-191`` sliced contiguously would
    have put two ranks on the two hyperthreads of the same cores;
  * the mask is applied to every thread the process already has (``/proc/self/task``): ``sched_setaffinity(0, ...)`` alone pins the calling thread.
Must run before the first HIP call of the process (bench.py, train.py call it right after parsing their arguments).  FD_NO_AFFINITY=1 leaves the
process alone.  A single-rank run is pinned to the whole NUMA node of its GPU when sysfs names it, with the same cap on the intra-op pool."""
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


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def numa_nodes(root="/"):
    """[[cpus of node 0], [cpus of node 1], ...] restricted to the CPUs this process may use; one pseudo-node when sysfs has none."""
    allowed = set(os.sched_getaffinity(0)) if root == "/" else None
    nodes = []
    for d in sorted(glob.glob(os.path.join(root, "sys/devices/system/node/node[0-9]*")), key=lambda p: int(p.rsplit("node", 1)[1])):
        text = _read(os.path.join(d, "cpulist"))
        if text is None:
            continue
        cpus = [c for c in _parse_cpulist(text) if allowed is None or c in allowed]
        nodes.append(cpus)          # keep empty nodes: the index is the NUMA node id the GPU's numa_node file refers to
    if not any(nodes):
        return [sorted(allowed if allowed is not None else [])]
    return nodes


def cores_of(cpus, root="/"):
    """The logical CPUs of ``cpus`` grouped by physical core, [[cpu, sibling, ...], ...] in ascending order of each core's first CPU."""
    left, cores = set(cpus), []
    for c in sorted(cpus):
        if c not in left:
            continue
        text = _read(os.path.join(root, f"sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list"))
        sib = [s for s in (_parse_cpulist(text) if text else [c]) if s in left] or [c]
        if c not in sib:
            sib = [c]
        left -= set(sib)
        cores.append(sorted(sib))
    return cores


def gpu_numa_node(device_index, root="/", env=None):
    """NUMA node of HIP device ``device_index`` from sysfs, or None when it cannot be determined (no KFD topology, numa_node = -1, a visible-devices
    variable that is not a plain index list)."""
    env = os.environ if env is None else env
    idx = device_index
    # HIP device i -> HIP_VISIBLE_DEVICES[i] (CUDA_VISIBLE_DEVICES is its alias on ROCm and is IGNORED when HIP_VISIBLE_DEVICES is set), an index into the
    # list the ROCr runtime exposes -> ROCR_VISIBLE_DEVICES[that] = the physical GPU (the three were chained in the wrong order, and HIP + CUDA
    # both set were applied twice)
    for v in (env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES"), env.get("ROCR_VISIBLE_DEVICES")):
        if v:
            try:
                idx = [int(x) for x in v.split(",")][idx]
            except (ValueError, IndexError):
                return None
    gpus = []
    for d in sorted(glob.glob(os.path.join(root, "sys/class/kfd/kfd/topology/nodes/[0-9]*")), key=lambda p: int(os.path.basename(p))):
        props = _read(os.path.join(d, "properties"))
        if props is None:
            continue
        kv = dict(line.split(None, 1) for line in props.splitlines() if len(line.split(None, 1)) == 2)
        if int(kv.get("simd_count", "0")) > 0 and int(kv.get("drm_render_minor", "-1")) >= 0:
            gpus.append(int(kv["drm_render_minor"]))
    if idx >= len(gpus):
        return None
    text = _read(os.path.join(root, f"sys/class/drm/renderD{gpus[idx]}/device/numa_node"))
    try:
        node = int(text)
    except (TypeError, ValueError):
        return None
    return node if node >= 0 else None


def rank_cpus(local_rank, local_world, nodes=None, gpu_nodes=None, root="/"):
    """The CPUs rank ``local_rank`` of ``local_world`` is pinned to (pure function of the node layout: testable without touching the process).
    ``gpu_nodes``: NUMA node of every local rank's device (None entries = unknown); default: read from sysfs."""
    nodes = numa_nodes(root) if nodes is None else nodes
    nn = len(nodes)
    if gpu_nodes is None:
        gpu_nodes = [gpu_numa_node(r, root) for r in range(local_world)]
    if any(g is None or g >= nn or not nodes[g] for g in gpu_nodes):
        # round 4's even deal -- over the nodes that HAVE CPUs (list indices are NUMA ids, so a cpuset confined to one socket leaves empty entries; dealing
        # ranks onto those left them unpinned with the machine-sized intra-op pool)
        ne = [i for i, n in enumerate(nodes) if n] or [0]
        gpu_nodes = [ne[min(r * len(ne) // local_world, len(ne) - 1)] for r in range(local_world)]
    node = gpu_nodes[local_rank]
    peers = [r for r in range(local_world) if gpu_nodes[r] == node]
    cores = cores_of(nodes[node], root)
    k, n = peers.index(local_rank), len(peers)
    per = max(len(cores) // n, 1)
    mine = cores[k * per:(k + 1) * per] if k < n - 1 else cores[k * per:]
    share = sorted(c for core in mine for c in core)
    return share or sorted(nodes[node])


def single_rank_cpus(device_index, root="/", env=None):
    """The CPUs a single-rank run is pinned to: the NUMA node of its GPU, or None when sysfs does not name one (or the host has one node)."""
    nodes = numa_nodes(root)
    g = gpu_numa_node(device_index, root, env)
    if g is None or g >= len(nodes) or not nodes[g] or len(nodes) < 2:
        return None
    return sorted(nodes[g])


def pin_rank(local_rank, local_world, max_threads=8):
    """Pins the process (all of its threads) and sizes torch's intra-op pool; returns (cpus, threads) or None when nothing was done."""
    if os.environ.get("FD_NO_AFFINITY") is not None or not hasattr(os, "sched_setaffinity"):
        return None
    single = local_world <= 1
    if single:
        # one rank: the whole NUMA node its GPU hangs off -- and only when sysfs names that node (a guess could pin the enqueue thread to the far socket).  Same box,
        # alternating runs of 12 steps (profiles/r05_single_rank_numa_pinning.txt): 1309.4 / 1309.5 / 1309.7 ms pinned against 1313.4 / 1318.0 / 1317.7 left alone, whose
        # extra is two of the +70 ms steps
        cpus = single_rank_cpus(local_rank)
        if cpus is None:
            return None
    else:
        cpus = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    for t in os.listdir("/proc/self/task") if os.path.isdir("/proc/self/task") else []:     # threads that exist already (BLAS / OpenMP pools started at import)
        try:
            os.sched_setaffinity(int(t), cpus)
        except (OSError, ValueError):
            pass
    import torch
    # the intra-op pool is capped for a single rank too: left at its default (128 threads on the pool's hosts, whose jobs get far fewer cores than CPUs) the pinned
    # run was 1405 ms per step against 1309 with 8 threads -- the pool's spinning workers take the enqueue thread's cycles (bench.py's CPU baseline sizes its own pool)
    threads = max(1, min(len(cpus), max_threads))
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    torch.set_num_threads(threads)
    return cpus, threads
