"""graphinvent_amd: the MI355X-native GGNN / APD-readout hot path of GraphINVENT."""
import os as _os
import sys as _sys

# hipGraph replays of this library's launches come out WRONG on ROCm 7.0.2 with the runtime's AQL-packet
# capture of graph kernel nodes enabled (its default): replays that follow other eager launches run some
# kernels with stale arguments (measured, round 3: a captured bounded forward replayed after three eager
# forwards returns logits off by O(1), deterministically; with the runtime flag below it is bit-identical to the
# eager forward, and so are 100 % of the replays of the -m gpu tests; profiles/r03/hipgraph_replay_runtime_flag.txt).
# The flag only matters to code that captures hipGraphs.  It is read when the HIP runtime initialises, i.e. at the
# first CUDA call of the process, and it is PROCESS-WIDE (every library in the process gets the runtime's slower
# node-by-node graph launch path): importing this package before anything touches the GPU is enough.  If that is
# not the case — CUDA already initialised (e.g. the drop-in `BlockDatasetLoader` / `gnn` modules imported after the
# caller's own CUDA work), or the caller exported the flag as 1 — nothing can be changed any more: the state is
# recorded here and the capture entry points (a sync-free forward under stream capture) REFUSE to record instead of
# replaying garbage; `graph_capture_safe()` / `assert_graph_safe()` let callers check up front.
_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _cuda_already_initialised() -> bool:
    torch = _sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:                                            # pragma: no cover
        return False


_late = _cuda_already_initialised()
_preset = _os.environ.get(_FLAG)
if _preset is None and not _late:
    _os.environ[_FLAG] = "0"
if _preset is not None and _preset.strip() != "0":
    _GRAPH_UNSAFE = f"{_FLAG}={_preset} was set by the caller (hipGraph replays of this library need 0)"
elif _preset is None and _late:
    _GRAPH_UNSAFE = (f"the HIP runtime was initialised before graphinvent_amd was imported, so {_FLAG}=0 could not "
                     "take effect (import graphinvent_amd — or export the flag — before the first CUDA call)")
else:
    _GRAPH_UNSAFE = None        # ours, or the caller's own 0 (if CUDA was up already we trust it was there in time)


def graph_capture_safe():
    """(ok, reason): may this process capture this library's launches into a hipGraph and trust the replays?"""
    return _GRAPH_UNSAFE is None, _GRAPH_UNSAFE


def assert_graph_safe() -> None:
    """Raise unless hipGraph capture of this library's launches is known to replay correctly in this process."""
    if _GRAPH_UNSAFE is not None:
        raise RuntimeError("graphinvent_amd: hipGraph capture refused — " + _GRAPH_UNSAFE +
                           "; replays would run kernels with stale arguments on ROCm 7.0.x (DESIGN.md §6)")
