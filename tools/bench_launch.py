"""bench.py --gpus N: everything in front of the ranks -- the refusal rules of a multi-GPU line, the device census without a
runtime, the preflight probe (child process) and the launcher that starts one rank per GPU as a child and relays rank 0's
line.  Nothing here touches the GPU in the calling process."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def wants_library_exchange(backend, one_device, torch_gather):
    """The record exchange of a real multi-GPU run goes through the library's own RCCL communicator (vsg_shard_*);
    torch.distributed carries it only in the rehearsals: gloo, --one-device, or an explicit --torch-gather."""
    return backend == "nccl" and not one_device and not torch_gather


def refuse_without_library_exchange(rank, world, why):
    """--gpus N > 1 on the nccl backend without --torch-gather and vsg_shard_create failed on some rank: exit non-zero with
    the reason, on every rank, instead of measuring torch's all-gather under this library's name."""
    msg = (f"[bench] rank {rank}/{world}: vsg_shard_create failed on at least one rank"
           + (f" (here: {why})" if why else " (not on this one)")
           + "; refusing to fall back to torch.distributed -- pass --torch-gather to measure that on purpose")
    print(msg, file=sys.stderr)
    raise SystemExit(3)


def check_rccl_world(seen, world, rank=0):
    """The multi-GPU line is printed only if the live communicator (ncclCommCount) spans exactly the launched ranks."""
    if seen != world:
        print(f"[bench] rank {rank}: RCCL communicator spans {seen} ranks, launched {world}: no line", file=sys.stderr)
        raise SystemExit(4)
    return True


def visible_gpu_count():
    """GPUs this process may use, WITHOUT importing torch or initialising HIP (the launcher below must not touch the GPU
    before it starts its ranks): the visibility variables if set, else the KFD topology (nodes with SIMDs); None when
    neither answers -- the ranks then report a missing device themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    try:
        for node in Path("/sys/class/kfd/kfd/topology/nodes").iterdir():
            props = dict(line.split()[:2] for line in (node / "properties").read_text().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n:
            return n
    except (OSError, ValueError):
        pass
    return None


def preflight(n, batch, workload, one_device=False, quiet=False):
    """`bench.py --gpus N --preflight`: what a first N-GPU run can trip over, checked in a CHILD process (it touches
    the GPU) before any rank is started -- visible devices; that libvsg_orb.so and torch resolve the SAME libamdhip64
    (bench.py hands torch streams and device pointers to a library that links the runtime by soname: INTEGRATION.md
    section 5); that every pair of the N devices is peer-accessible (hipDeviceCanAccessPeer) and over which link
    (hipExtGetLinkTypeAndHopCount: xgmi / pcie); that the RCCL the library will dlopen exports what vsg_shard_* binds,
    ncclCommCount included; the record exchange's receive buffer (world x batch records per rank) against the free memory of every device.  Prints one JSON
    object; exit code 0 only when every check passed."""
    import subprocess
    code = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, %r)
n, batch, workload, one_device = %d, %d, %r, %r
out = {"requested_gpus": n, "checks": {}}
def check(name, ok, **info):
    out["checks"][name] = dict(ok=bool(ok), **info)
class DlInfo(C.Structure):
    _fields_ = [("fname", C.c_char_p), ("fbase", C.c_void_p), ("sname", C.c_char_p), ("saddr", C.c_void_p)]
libdl = C.CDLL(None)
libdl.dladdr.argtypes = [C.c_void_p, C.POINTER(DlInfo)]
def owner(lib, sym):
    addr = C.cast(getattr(lib, sym), C.c_void_p).value
    info = DlInfo()
    libdl.dladdr(addr, C.byref(info))
    return addr, (info.fname or b"?").decode()
import torch
from visual_sgraphs_amd import orb, sharding
import bench
L = orb.load_library()
tlib = None
tdir = os.path.join(os.path.dirname(torch.__file__), "lib")
for cand in ("libtorch_hip.so", "libc10_hip.so"):
    try:
        tlib = C.CDLL(os.path.join(tdir, cand)); break
    except OSError:
        pass
try:
    a_vsg, f_vsg = owner(L, "hipMalloc")
    a_t, f_t = owner(tlib, "hipMalloc") if tlib is not None else (None, "torch's HIP library not found")
    check("one_hip_runtime", a_vsg == a_t and os.path.realpath(f_vsg) == os.path.realpath(f_t), libvsg_orb=f_vsg, torch=f_t)
except Exception as e:
    check("one_hip_runtime", False, error=repr(e))
ndev_t = torch.cuda.device_count()
ndev_v = L.vsg_device_count()
need_dev = 1 if one_device else n  # --one-device: every rank on device 0 (dry runs of the multi-rank path)
check("devices", ndev_t >= need_dev and ndev_v >= need_dev, torch_device_count=ndev_t, vsg_device_count=ndev_v,
      visible_without_runtime=bench.visible_gpu_count(), needed=need_dev)
# every pair of the N devices peer-accessible, and over which link (round 6, VERDICT r5 #5c): RCCL's all-gather and the
# neighbour send / recv go device to device; a pair without peer access would fall back to staging through the host
try:
    hip = C.CDLL(f_vsg)  # the libamdhip64 the process already runs on
    hip.hipDeviceCanAccessPeer.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
    hip.hipExtGetLinkTypeAndHopCount.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    LINK = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi"}
    pairs, bad = [], []
    ndev_p = 0 if one_device else min(n, ndev_t)
    for a in range(ndev_p):
        for b in range(ndev_p):
            if a == b:
                continue
            can, lt, hops = C.c_int(0), C.c_uint32(99), C.c_uint32(0)
            rc1 = hip.hipDeviceCanAccessPeer(C.byref(can), a, b)
            rc2 = hip.hipExtGetLinkTypeAndHopCount(a, b, C.byref(lt), C.byref(hops))
            pairs.append({"from": a, "to": b, "peer": bool(can.value) and rc1 == 0,
                          "link": LINK.get(lt.value, str(lt.value)) if rc2 == 0 else None, "hops": hops.value if rc2 == 0 else None})
            if rc1 != 0 or not can.value:
                bad.append((a, b))
    links = sorted({p_["link"] for p_ in pairs if p_["link"]})
    check("peer_access", not bad, device_pairs_checked=len(pairs), pairs_without_peer_access=bad, link_types=links,
          all_xgmi_one_hop=bool(pairs) and all(p_["link"] == "xgmi" and p_["hops"] == 1 for p_ in pairs),
          note="one device per rank; --one-device and N = 1 have no pairs to check")
except Exception as e:
    check("peer_access", False, error=repr(e))
rccl = None
for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
    try:
        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL); break
    except OSError:
        pass
need = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart",
        "ncclGroupEnd", "ncclGetErrorString", "ncclCommCount", "ncclCommUserRank"]
if rccl is None:
    check("rccl", False, error="librccl.so.1 not loadable")
else:
    missing = [x for x in need if not hasattr(rccl, x)]
    _, f_r = owner(rccl, "ncclCommCount") if not missing else (None, "?")
    uid = (C.c_uint8 * 128)()
    rc = L.vsg_shard_unique_id(uid)
    check("rccl", not missing and rc == 0, library=f_r, missing=missing, vsg_shard_unique_id=rc)
W, H, nfeat = bench.WORKLOADS[workload]
cap = nfeat + 3 * 8  # >= vsg_orb_capacity for 8 levels; the handle's own figure needs a device allocation
rec = sharding.record_bytes(cap + 64)
recv = n * batch * rec
resident = batch * (W * H * 2 + int(2.6 * 1.38 * W * H) + 2 * (cap + 64) * 60 + 3 * (cap + 64) * 4)
mem = []
for d in range(min(need_dev, ndev_t)):
    free, total = torch.cuda.mem_get_info(d)
    mem.append({"device": d, "free_bytes": free, "total_bytes": total})
ranks_per_dev = n if one_device else 1
check("memory", bool(mem) and all(m["free_bytes"] > 2 * ranks_per_dev * (recv + resident) for m in mem), exchange_recv_bytes_per_rank=recv,
      resident_estimate_bytes_per_rank=resident, devices=mem)
out["ok"] = all(c["ok"] for c in out["checks"].values())
print(json.dumps(out))
sys.exit(0 if out["ok"] else 4)
""" % (str(ROOT), n, batch, workload, bool(one_device))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(ROOT))
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if lines and not quiet:
        print(lines[-1], flush=True)
    elif lines and r.returncode != 0:
        sys.stderr.write(lines[-1] + "\n")
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        if lines:
            bad = [k for k, v in json.loads(lines[-1])["checks"].items() if not v["ok"]]
            sys.stderr.write(f"bench.py --preflight: FAILED checks: {bad}\n")
        else:
            sys.stderr.write(f"bench.py --preflight: the probe died (exit code {r.returncode})\n")
    return r.returncode


def run_teeing_stderr(cmd, env):
    """Run `cmd`, relaying its stderr LIVE (a multi-GPU run that hangs in the RCCL rendezvous or in IPC shows its
    diagnostics while it hangs) and keeping a copy for the caller; stdout is captured (the one JSON line)."""
    import subprocess
    import threading
    import types
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kept = []

    def pump():
        for line in p.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            kept.append(line)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    out = p.stdout.read()
    p.wait()
    t.join()
    return types.SimpleNamespace(returncode=p.returncode, stdout=out, stderr="".join(kept[-400:]))


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): start the N ranks -- one process per GPU,
    `python -m torch.distributed.run` -- as a CHILD process before this one imports torch or touches a GPU, relay rank
    0's JSON line, and fail loudly (non-zero exit, no line) if fewer than N devices are visible, a rank fails, or the line
    does not say n_gpus == N.  Never a silent one-rank line for an N-GPU request."""
    import socket
    import subprocess
    n = args.gpus
    if not args.one_device:
        have = visible_gpu_count()
        if have is not None and have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible on this node\n")
            return 2
    if not args.no_preflight:
        # what an N-GPU run depends on, checked in a child BEFORE any rank starts (stdout stays the one result line): a
        # torch wheel with another libamdhip64 than libvsg_orb.so resolves, an RCCL without ncclCommCount, too little memory
        # for the exchange buffers -- each ends here with its reason instead of inside a hung or crashed rank
        rc = preflight(n, args.batch, args.workload, one_device=args.one_device, quiet=True)
        if rc != 0:
            sys.stderr.write(f"bench.py: --gpus {n}: preflight failed (exit code {rc}); no ranks started (--no-preflight skips it)\n")
            return rc
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = None
    for attempt in range(3):
        # a free port is found by bind-then-close, so another process may take it before torchrun binds it: a run
        # that dies on the rendezvous address is started again on another port
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py")] + list(argv)
        r = run_teeing_stderr(cmd, env)
        # started again only when the RENDEZVOUS lost its port: the address error is there and no rank got as far as
        # printing anything of its own (a rank-side socket error of gloo / RCCL carries the same words and is a real failure)
        addr_in_use = any(m in r.stderr for m in ("EADDRINUSE", "Address already in use", "address already in use"))
        if r.returncode == 0 or not addr_in_use or "[bench]" in r.stderr or r.stdout.strip():
            break
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    for x in r.stdout.splitlines():
        if not x.startswith("{"):
            sys.stderr.write(x + "\n")
    if r.returncode != 0:
        sys.stderr.write(f"bench.py: the {n}-rank run failed (exit code {r.returncode}); no result line\n")
        return r.returncode
    try:
        d = json.loads(lines[-1])
    except (IndexError, ValueError):
        sys.stderr.write("bench.py: the ranks printed no JSON line\n")
        return 3
    if d.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: asked for {n} GPUs, the line says n_gpus = {d.get('n_gpus')}\n")
        return 3
    print(lines[-1], flush=True)
    return 0
