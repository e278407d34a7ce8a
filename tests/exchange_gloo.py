"""Test helper (not product code: the product merges inside the library, kamd_ec_allreduce): the multi-GPU merge of the EC state written
with torch.distributed, as the CPU gloo test of the merge logic uses it.

The reference merges per-thread results under a mutex: `tc.counts[i] += c[i]` plus transfer of newly discovered ECs
(MasterProcessor::update, src/ProcessReads.cpp:424-499).  Here every rank holds
  (a) a dense count vector over the index's transcript sets, and
  (b) records [count, m, e0..e(m-1)] for read pairs whose hits carried m > 1 index sets,
both keyed by index set ids that are identical on every rank.  Merge = ONE all-reduce(sum) of (a) + an all-gather of
(b); the gathered records are then de-duplicated by the same kernels that de-duplicate a single rank's records.
This module is device-agnostic torch code (it runs on CPU tensors over gloo in the tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def merge_ec_state(dense: torch.Tensor, words: torch.Tensor, offs: torch.Tensor, group=None):
    """dense: int32 [n_index_sets] (summed in place); words: int32 record words; offs: int64 word offset of each record.
    Returns (words_all, offs_all) -- the records of every rank concatenated in rank order, offsets rebased."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return words, offs
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    return gather_records(words, offs, group)


def _via_host(t: torch.Tensor, group) -> bool:
    """gloo has no all_gather for device tensors: stage through the host (tests / single-GPU smoke runs only)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_records(words: torch.Tensor, offs: torch.Tensor, group=None):
    """All-gather variable-length record buffers: (words, offsets) of every rank concatenated, offsets rebased."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return words, offs
    world = dist.get_world_size(group)
    if _via_host(words, group):
        dev = words.device
        w, o = gather_records(words.cpu(), offs.cpu(), group)
        return w.to(dev), o.to(dev)
    dense = words
    sizes = torch.tensor([words.numel(), offs.numel()], dtype=torch.int64, device=dense.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu()
    mw, mo = max(int(all_sizes[:, 0].max()), 1), max(int(all_sizes[:, 1].max()), 1)
    pw = torch.zeros(mw, dtype=torch.int32, device=dense.device)
    po = torch.zeros(mo, dtype=torch.int64, device=dense.device)
    pw[:words.numel()] = words
    po[:offs.numel()] = offs
    gw = [torch.empty_like(pw) for _ in range(world)]
    go = [torch.empty_like(po) for _ in range(world)]
    dist.all_gather(gw, pw, group=group)
    dist.all_gather(go, po, group=group)
    cat_w, cat_o, base = [], [], 0
    for r in range(world):
        nw, no = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        cat_w.append(gw[r][:nw])
        cat_o.append(go[r][:no] + base)
        base += nw
    return torch.cat(cat_w).contiguous(), torch.cat(cat_o).contiguous()
