"""Host-side pieces added in round 5 (no GPU): output slabs, the device table builder's eligibility rule, the rendezvous
store selection of the RCCL communicator, the mixed workload's fabricated records."""
import os
import sys

import numpy as np
import torch

from xenoverse_amd.anymdp.tables import device_buildable
from xenoverse_amd.vector import OutputSlabs


def test_output_slabs_hand_out_disjoint_tensors_with_matching_pointers():
    fields = [("obs", torch.int32, ()), ("steps", torch.int32, ()), ("reward", torch.float32, ()), ("frame", torch.float32, (3, 2)),
              ("term", torch.uint8, ()), ("done", torch.uint8, ())]
    sl = OutputSlabs(fields, 7, "cpu", K=4, as_bool=("term", "done"), order=("obs", "reward", "term", "frame"))
    seen, kept = set(), []
    for step in range(11):                       # three slabs
        t, p = sl.next()
        assert [x.value for x in p] == [t[k].data_ptr() for k in ("obs", "reward", "term", "frame")]
        assert t["obs"].shape == (7,) and t["frame"].shape == (7, 3, 2) and t["term"].dtype == torch.bool
        for k, v in t.items():
            lo = v.data_ptr()
            span = (lo, lo + v.numel() * v.element_size())
            for a, b in seen:                    # no tensor overlaps any tensor handed out before
                assert span[1] <= a or span[0] >= b, (step, k)
            seen.add(span)
            v.fill_(1 if v.dtype == torch.bool else step + 1)
        kept.append(t)
    for step, t in enumerate(kept):              # what a step got is never written again
        assert int(t["obs"][0]) == step + 1 and float(t["frame"][6, 2, 1]) == step + 1 and bool(t["done"][3])


def test_output_slabs_are_recycled_only_when_nobody_can_reach_them():
    """round 6: a used-up slab is handed out again — the same tensor objects — once no Python reference, no C++ holder (DLPack)
    and no alias of its storage (view, detach, numpy) is left; anything the caller still holds keeps its values"""
    import torch.utils.dlpack as dl
    fields = [("obs", torch.int32, ()), ("reward", torch.float32, ()), ("term", torch.uint8, ())]
    sl = OutputSlabs(fields, 5, "cpu", K=4, as_bool=("term",), order=("obs", "reward", "term"))
    last = None
    ids = set()
    for step in range(48):                       # the usual loop: only the latest observation is kept
        t, p = sl.next()
        t["obs"].fill_(step)
        last = t["obs"]
        ids.add(id(t["obs"]))
    assert sl.made == 2 and sl.recycled == 10 and len(ids) == 8      # two slabs alternate: no new objects after the first two
    assert int(last[0]) == 47
    holders = {}
    for step in range(64):
        t, p = sl.next()
        for k, v in t.items():
            v.fill_(1 if v.dtype == torch.bool else step + 100)
        if step == 3:
            holders["tensor"] = (t["reward"], lambda x: float(x[0]), 103.0)
        if step == 9:
            holders["view"] = (t["obs"][1:3], lambda x: int(x[0]), 109)
        if step == 14:
            holders["detach"] = (t["reward"].detach(), lambda x: float(x[4]), 114.0)
        if step == 21:
            holders["numpy"] = (t["obs"].numpy(), lambda x: int(x[2]), 121)
        if step == 26:
            holders["dlpack"] = (dl.to_dlpack(t["obs"]), None, 126)
    for _ in range(64):                          # plenty of further steps: every free slab is recycled and overwritten
        t, p = sl.next()
        for v in t.values():
            v.fill_(0)
    for name, (h, get, want) in holders.items():
        if name == "dlpack":
            h = dl.from_dlpack(h)
            get = lambda x: int(x[0])           # noqa: E731
        assert get(h) == want, name
    made = sl.made
    holders.clear()
    del h
    for _ in range(64):
        sl.next()
    assert sl.made == made                       # once the holders are gone their slabs serve again: nothing new is allocated


def test_device_buildable_rule():
    def task(n, a, extra=None):
        d = dict(transition=np.zeros((n, a, n)), reward=np.zeros((n, a, n)), reward_noise=np.zeros((n, a, n)), na=a)
        d.update(extra or {})
        return d
    assert device_buildable([task(8, 3), task(8, 3)])
    assert not device_buildable([task(8, 3), task(9, 3)])            # ragged: host builder pads
    assert not device_buildable([task(1, 3)])                        # bandits are embedded by the host builder
    assert not device_buildable([task(8, 3, {"na": 4})])
    assert not device_buildable(task(8, 3)) and not device_buildable([])
    assert not device_buildable([task(8, 3, {"reward": np.zeros((8, 3, 7))})])


def test_rendezvous_store_prefers_the_callers_store():
    from xenoverse_amd.distributed import _rendezvous_store

    class Store(dict):
        def set(self, k, v):
            self[k] = v

        def get(self, k):
            return self[k]
    st = Store()
    assert _rendezvous_store(0, 2, st, None, None, 5) is st


def test_mixed_workload_fabricated_records_are_functions_of_global_ids():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_mixed
    from xenoverse_amd.distributed import MixedChunk
    ch = MixedChunk(4, 256, 128, 64, 2)
    whole = bench_mixed.fabricate(torch, ch, 4, {"anymdp": (0, 256), "linds": (0, 128), "cartpole": (0, 64)})
    for r in range(2):
        part = bench_mixed.fabricate(torch, ch, 4, {f: ch.share[f][r] for f in ch.rec})
        for f in ch.rec:
            lo, hi = ch.share[f][r]
            for k in part[f]:
                assert torch.equal(part[f][k], whole[f][k][:, lo:hi]), (f, k)
    assert ch.bytes_per_rank == 4 * (128 * 8 + 64 * 72 + 32 * 24)


def test_profile_hash_covers_the_kernel_section_only(tmp_path, monkeypatch):
    """xenoverse_amd.build.source_hash ties a committed counter profile to the KERNEL source it measured: a .hip file counts
    up to its `#ifndef XV_KERNELS_ONLY` line, headers whole — host-side edits behind the marker leave the hash alone"""
    from xenoverse_amd import build as xb
    d = tmp_path / "csrc"
    d.mkdir()
    (d / "k.hip").write_text("__global__ void k() {}\n#ifndef XV_KERNELS_ONLY\nint host_a;\n#endif\n")
    (d / "h.h").write_text("#define X 1\n")
    monkeypatch.setattr(xb, "CSRC", str(d))
    h0 = xb.source_hash(("k.hip", "h.h"))
    (d / "k.hip").write_text("__global__ void k() {}\n#ifndef XV_KERNELS_ONLY\nint host_b; int more;\n#endif\n")
    assert xb.source_hash(("k.hip", "h.h")) == h0                      # host part changed: same hash
    (d / "k.hip").write_text("__global__ void k() { }\n#ifndef XV_KERNELS_ONLY\nint host_b;\n#endif\n")
    assert xb.source_hash(("k.hip", "h.h")) != h0                      # kernel part changed
    (d / "k.hip").write_text("__global__ void k() {}\n#ifndef XV_KERNELS_ONLY\nint host_a;\n#endif\n")
    (d / "h.h").write_text("#define X 2\n")
    assert xb.source_hash(("k.hip", "h.h")) != h0                      # a header counts whole
    # the committed profile of the step kernel is keyed to the source in the tree
    monkeypatch.undo()
    import json, glob, os
    import bench
    prof = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r0*pmc_traffic_anymdp_2a*.json")))
    assert prof and json.load(open(prof[-1]))["bench_key"]["kernel_source_sha16"] == bench.kernel_source_hash()
