"""Checkpoint files -> {tensor name: ndarray} for the band scripts.

The reference loads its models with torch (`torch.load` + `load_state_dict`): RAFT keeps DataParallel's `module.` prefix
(bands/flow_raft.py:38-46), GMFlow wraps the state dict as `{'model': ...}` (bands/flow_gmflow.py:57-61), mmdet as
`{'meta': ..., 'state_dict': ...}` (bands/mmdet/apis/inference.py init_detector), ZoeDepth as `{'model': ...}`, Depth-Anything ships a
plain state dict.  The engines take float32 ndarrays keyed by the reference's parameter names; BatchNorm's `num_batches_tracked`
(int64) and any other non-floating entry is kept as it is and ignored by them.  `.npz` files (np.savez of such a dict) are accepted
wherever a `.pth` is.  torch is imported only when a torch checkpoint is actually read.
"""
import numpy as np


def _to_numpy(v):
    """fp16 / bf16 / fp64 checkpoints become float32 (bf16 has no numpy dtype); integer buffers stay integers."""
    if hasattr(v, "detach"):                    # torch.Tensor (or Parameter)
        v = v.detach().cpu()
        return v.float().numpy() if v.is_floating_point() else v.numpy()
    a = np.asarray(v)
    return a.astype(np.float32) if a.dtype.kind == "f" and a.dtype != np.float32 else a


def load_checkpoint(path, wrappers=(), strip_prefix=""):
    """wrappers: keys under which the state dict may sit (the first one present is unwrapped); strip_prefix: removed from the names
    that carry it.  Entries that are not tensors / arrays (mmdet's `meta`, optimizer state left beside a state dict) are dropped."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            sd = {k: z[k] for k in z.files}
    else:
        import pickle
        import sys
        import torch
        # tensors-only unpickling first (RAFT, GMFlow, Depth-Anything, ZoeDepth checkpoints are plain tensor dicts): a .pth from an
        # untrusted source then cannot run code.  mmdet's `meta` entry holds arbitrary Python objects and needs the full unpickler -
        # taken only when the restricted one refuses the file, and said on stderr
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except TypeError:                       # torch < 1.13 has no weights_only
            sd = torch.load(path, map_location="cpu")
        except (pickle.UnpicklingError, RuntimeError) as e:
            print(f"[prisma] {path}: not a tensors-only checkpoint ({str(e).splitlines()[0][:120]}); loading it with the full unpickler - "
                  "only do this with files you trust", file=sys.stderr)
            sd = torch.load(path, map_location="cpu", weights_only=False)
    for w in wrappers:
        if isinstance(sd, dict) and w in sd and isinstance(sd[w], dict):
            sd = sd[w]
            break
    if not isinstance(sd, dict):
        raise ValueError(f"{path}: expected a state dict (optionally under one of {list(wrappers)}), found a pickled {type(sd).__name__} - "
                         "save `model.state_dict()`, not the module")
    out = {}
    for k, v in sd.items():
        if not (hasattr(v, "detach") or isinstance(v, np.ndarray)):
            continue
        name = k[len(strip_prefix):] if strip_prefix and k.startswith(strip_prefix) else k
        out[name] = _to_numpy(v)
    return out
