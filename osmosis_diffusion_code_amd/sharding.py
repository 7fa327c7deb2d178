"""Multi-GPU = independent image chains: rank r of W owns images[r::W]; UNet
weights are replicated; there is NO collective on the data path.  The only cross-rank traffic is
the end-of-run bookkeeping below (a MAX over per-rank wall times and a gather of per-image
results), which works on any torch.distributed backend (RCCL on GPUs, gloo in the CPU tests)."""
from typing import List, Sequence

import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    return list(range(n_items))[rank::world]


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a scalar over all ranks (identity when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() == "gloo":
        device = None                    # gloo reduces host tensors
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_image(values: Sequence[float], n_items: int, device=None) -> List[float]:
    """Every rank passes the values of ITS images (in shard order); returns the full list in image order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo":
        device = None
    per = (n_items + world - 1) // world
    mine = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
    mine[:len(values)] = torch.tensor(list(values), dtype=torch.float64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    out = [float("nan")] * n_items
    for r in range(world):
        for k, idx in enumerate(shard_indices(n_items, r, world)):
            out[idx] = float(allv[r][k])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Rank bookkeeping that cannot lose a run.  The data path needs no collective (images are independent chains), so a
# broken collective library must never cost the measurement: `RankSync` offers barrier / all-gather-of-floats over the
# first transport that WORKS ON EVERY RANK, tried in this order:
#   "rccl"  -- torch.distributed backend "nccl" (= RCCL on ROCm), device tensors, one rank per GPU over xGMI;
#   "gloo"  -- torch.distributed backend "gloo", host tensors over TCP on 127.0.0.1;
#   "files" -- one small JSON file per rank and round in a node-local directory (the contract is ONE node), polled.
# The directory may be REUSED (OSM_SYNC_DIR, or a (run id, port, parent pid) collision): rank 0 empties it and then publishes a
# job token {its pid, that process's kernel start time, a random nonce}; a peer accepts a token only while that very process
# is alive (one node: /proc), every payload carries the nonce and a file with another nonce -- a previous job's -- is ignored
# (a stale vote_* / g<N> file was accepted as this run's).  close() removes what the job wrote.
# Agreement on the transport itself goes through that directory, so every rank takes the same branch even when the failure
# is one-sided.  A transport whose probe raises OR does not finish in `probe_timeout_s` (a hang) counts as failed; a
# hung RCCL probe thread is abandoned (daemon) and `device_sync_safe` turns False: callers then synchronise their own
# stream instead of the whole device (a wedged RCCL kernel would block `torch.cuda.synchronize()`).
class RankSync:
    ORDER = ("rccl", "gloo", "files")
    _instances = 0              # RankSync objects this process has built (SPMD: the same number on every rank): part of the job token

    def __init__(self, rank: int, world: int, device=None, sync_dir: str = None, probe_timeout_s: float = 120.0,
                 force_fail: Sequence[str] = ()):
        import os
        import sys
        import time
        self.rank, self.world, self.device = int(rank), int(world), device
        self.transport, self.failures = "none", {}
        self.device_sync_safe = True
        self.setup_s = 0.0
        RankSync._instances += 1
        self._seq = RankSync._instances
        self._round = 0
        self._mine = []             # files this rank wrote (removed by close())
        self._group = None
        self._dist_up = False
        if self.world == 1:
            return
        if sync_dir is None:
            sync_dir = os.environ.get("OSM_SYNC_DIR") or os.path.join(
                "/tmp", "osm_sync_%s_%s_%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                                os.environ.get("MASTER_PORT", "0"), os.getppid()))
        os.makedirs(sync_dir, exist_ok=True)
        self.dir = sync_dir
        self._timeout = float(probe_timeout_s)
        t_all = time.monotonic()

        def say(msg):           # a broken node must not look like silence (up to probe_timeout_s per transport): stderr, at once
            print("[RankSync rank %d/%d] %s" % (self.rank, self.world, msg), file=sys.stderr, flush=True)
        self._nonce = self._agree_on_nonce()
        force_fail = set(force_fail) | set(filter(None, os.environ.get("OSM_SYNC_FORCE_FAIL", "").split(",")))
        for name in self.ORDER:
            if name == "files":
                self.transport = "files"
                break
            t0 = time.monotonic()
            ok, why = (False, "forced failure (test)") if name in force_fail else self._probe(name)
            say("probe %s: %s in %.1f s%s" % (name, "ok" if ok else "FAILED", time.monotonic() - t0, "" if ok else " (%s)" % why))
            votes = self._file_gather("vote_" + name, [1.0 if ok else 0.0])
            if not ok:
                self.failures[name] = why
            if all(v[0] == 1.0 for v in votes):
                self.transport = name
                break
            if ok:
                self.failures[name] = "failed on rank(s) %s" % [r for r, v in enumerate(votes) if v[0] != 1.0]
        self.setup_s = time.monotonic() - t_all
        say("transport = %s, device = %s, set up in %.1f s" % (self.transport, self.device, self.setup_s))

    @property
    def round(self) -> int:
        """Rounds (barriers + gathers) this rank has entered; equal on all ranks between two primitives."""
        return self._round

    # -- job token ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _proc_start(pid: int):
        """Kernel start time (clock ticks since boot, /proc/<pid>/stat field 22) of a live process, None if it is gone."""
        try:
            with open("/proc/%d/stat" % pid) as f:
                return int(f.read().rsplit(")", 1)[1].split()[19])
        except (OSError, ValueError, IndexError):
            return None

    def _agree_on_nonce(self) -> str:
        import json
        import os
        import time
        import uuid
        path = os.path.join(self.dir, "token_r0.json")
        if self.rank == 0:
            for f in os.listdir(self.dir):          # whatever an earlier job left behind (its ranks are gone: see below)
                try:
                    os.remove(os.path.join(self.dir, f))
                except OSError:
                    pass
            tok = {"pid": os.getpid(), "start": self._proc_start(os.getpid()), "born": time.time(), "nonce": uuid.uuid4().hex,
                   "seq": self._seq}
            with open(path + ".tmp", "w") as f:
                json.dump(tok, f)
            os.replace(path + ".tmp", path)
            return tok["nonce"]
        t0, limit = time.monotonic(), max(600.0, 4 * self._timeout)
        my_born = time.time()
        while True:
            try:
                with open(path) as f:
                    tok = json.load(f)
                # this job's token is the one whose writer is still running (pid + kernel start time identify a process on
                # the node); without /proc: a token not older than this process by more than two minutes
                # ... and that belongs to THIS RankSync of that process (a second RankSync in the same processes and directory must
                # not adopt the first one's token before rank 0 has replaced it: ADVICE r05)
                alive = self._proc_start(int(tok["pid"]))
                if int(tok.get("seq", self._seq)) == self._seq and (
                        (alive is not None and alive == tok["start"]) or (tok["start"] is None and tok["born"] > my_born - 120.0)):
                    return str(tok["nonce"])
            except (OSError, ValueError, KeyError, TypeError):
                pass
            if time.monotonic() - t0 > limit:
                raise RuntimeError("RankSync: rank 0 never published a job token in %s" % self.dir)
            time.sleep(0.001)

    # -- probes -------------------------------------------------------------------------------------------------------
    def _run_with_timeout(self, fn):
        import threading
        box = {}

        def body():
            try:
                fn()
                box["ok"] = True
            except BaseException as e:     # noqa: BLE001 -- any failure of a probe means "do not use this transport"
                box["err"] = "%s: %s" % (type(e).__name__, str(e).replace("\n", " ")[:300])

        th = threading.Thread(target=body, daemon=True)
        th.start()
        th.join(self._timeout)
        if th.is_alive():
            return False, "no answer within %.0f s (hang)" % self._timeout, True
        return ("ok" in box), box.get("err", ""), False

    def _probe(self, name):
        import datetime
        import os

        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=max(10.0, self._timeout))

        def ensure_default_group():
            # gloo is the default group: it only needs TCP on 127.0.0.1; RCCL rides on it as a sub-group, so a failed /
            # hung RCCL leaves a working process group behind
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=tmo)
            self._dist_up = True

        if name == "gloo":
            def fn():
                ensure_default_group()
                t = torch.tensor([float(self.rank)], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                assert float(t) == self.world * (self.world - 1) / 2.0, "gloo all_reduce returned a wrong sum"
            ok, why, _hung = self._run_with_timeout(fn)
            return ok, why

        def fn():
            assert self.device is not None and torch.device(self.device).type == "cuda", "rccl needs a HIP device"
            # a failed / timed-out RCCL collective must raise here, not let torch's NCCL watchdog tear the process down
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
            torch.cuda.set_device(self.device)          # the current device is per THREAD: this probe runs in its own
            ensure_default_group()
            g = dist.new_group(backend="nccl", timeout=tmo)
            t = torch.tensor([float(self.rank)], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=g)
            torch.cuda.current_stream(self.device).synchronize()
            assert float(t.item()) == self.world * (self.world - 1) / 2.0, "rccl all_reduce returned a wrong sum"
            dist.barrier(group=g, device_ids=[torch.device(self.device).index])
            self._group = g
        ok, why, hung = self._run_with_timeout(fn)
        if hung:
            self.device_sync_safe = False
        if not ok:
            self._group = None
        return ok, why

    # -- file transport (also the agreement channel) ---------------------------------------------------------------------
    def _file_gather(self, tag: str, values, timeout_s: float = None):
        import json
        import os
        import time
        self._file_post(tag, values)               # atomic: a reader sees the whole file or none
        out, t0 = [None] * self.world, time.monotonic()
        limit = timeout_s if timeout_s is not None else max(600.0, 4 * self._timeout)
        while True:
            for r in range(self.world):
                if out[r] is None:
                    try:
                        with open(os.path.join(self.dir, "%s_r%d.json" % (tag, r))) as f:
                            rec = json.load(f)
                        if isinstance(rec, dict) and rec.get("n") == self._nonce:     # another nonce: a previous job's file
                            out[r] = rec["v"]
                    except (FileNotFoundError, ValueError, KeyError):
                        pass
            if all(o is not None for o in out):
                return out
            if time.monotonic() - t0 > limit:
                missing = [r for r in range(self.world) if out[r] is None]
                raise RuntimeError("RankSync(files): ranks %s never wrote %r in %s" % (missing, tag, self.dir))
            time.sleep(0.0002)

    # -- the two primitives -----------------------------------------------------------------------------------------------
    # Every round ALSO leaves each rank's payload in the node-local directory before the collective is tried: a rank whose
    # collective raises (alone or with the others) finishes the round from those files, while the ranks whose collective
    # returned move on -- no rank can be left waiting for a fallback the others never entered.
    def _degrade(self, why: Exception):
        self.failures[self.transport] = "failed after selection: %s: %s" % (type(why).__name__, str(why).replace("\n", " ")[:200])
        self.transport = "files"

    def _file_post(self, tag: str, values):
        import json
        import os
        path = os.path.join(self.dir, "%s_r%d.json" % (tag, self.rank))
        with open(path + ".tmp", "w") as f:
            json.dump({"n": self._nonce, "v": [float(v) for v in values]}, f)
        os.replace(path + ".tmp", path)
        self._mine.append(path)

    def all_gather(self, values: Sequence[float]) -> List[List[float]]:
        """Every rank passes the same number of floats; returns the per-rank lists in rank order."""
        vals = [float(v) for v in values]
        if self.world == 1:
            return [vals]
        self._round += 1
        tag = "g%d" % self._round
        if self.transport == "files":
            return self._file_gather(tag, vals)
        self._file_post(tag, vals)
        try:
            import torch.distributed as dist
            dev = self.device if self.transport == "rccl" else None
            mine = torch.tensor(vals, dtype=torch.float64, device=dev)
            allv = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(allv, mine, group=self._group if self.transport == "rccl" else None)
            return [[float(x) for x in t.cpu()] for t in allv]
        except Exception as e:      # noqa: BLE001 -- the measurement outlives the collective library
            self._degrade(e)
            return self._file_gather(tag, vals)

    def barrier(self) -> None:
        if self.world == 1:
            return
        self._round += 1
        tag = "b%d" % self._round
        if self.transport == "files":
            self._file_gather(tag, [0.0])
            return
        self._file_post(tag, [0.0])
        try:
            import torch.distributed as dist
            if self.transport == "rccl":
                dist.barrier(group=self._group, device_ids=[torch.device(self.device).index])
                torch.cuda.current_stream(self.device).synchronize()   # an RCCL barrier is a kernel: the host must see it done
            else:
                dist.barrier()
        except Exception as e:      # noqa: BLE001
            self._degrade(e)
            self._file_gather(tag, [0.0])

    def max(self, value: float) -> float:
        return max(v[0] for v in self.all_gather([value]))

    def device_synchronize(self) -> None:
        if self.device is None or torch.device(self.device).type != "cuda":
            return
        if self.device_sync_safe:
            torch.cuda.synchronize(self.device)
        else:
            torch.cuda.current_stream(self.device).synchronize()

    def close(self) -> None:
        if self.world == 1:
            return
        import os
        # Files of every round but the LAST are removed: a peer has read round k's files before it could enter round k + 1, but a
        # slower peer may still be polling for this rank's file of the last round (removing it made that peer wait for its timeout:
        # seen once in the GPU suite).  The last round's few bytes stay until the next job's rank 0 empties the directory.
        keep = ("b%d_r" % self._round, "g%d_r" % self._round)
        for path in self._mine:
            if os.path.basename(path).startswith(keep):
                continue
            try:
                os.remove(path)
            except OSError:
                pass
        self._mine = []
        if self.rank == 0:          # the job token goes with the job: a later RankSync in this directory waits for ITS rank 0's token
            try:
                os.remove(os.path.join(self.dir, "token_r0.json"))
            except OSError:
                pass
        if not self._dist_up:
            return
        import torch.distributed as dist
        try:
            if self.device_sync_safe and dist.is_initialized():
                dist.destroy_process_group()
        except Exception:      # noqa: BLE001 -- teardown must not turn a finished measurement into a failure
            pass
