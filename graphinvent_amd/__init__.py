"""graphinvent_amd: the MI355X-native GGNN / APD-readout hot path of GraphINVENT."""
import os as _os

# hipGraph replays of this library's launches come out WRONG on ROCm 7.0.2 with the runtime's AQL-packet
# capture of graph kernel nodes enabled (its default): replays that follow other eager launches run some
# kernels with stale arguments (measured, round 3: tools/runs/dbg_graph8.py — a captured bounded forward
# replayed after three eager forwards returns logits off by O(1), deterministically; with the runtime flag
# below it is bit-identical to the eager forward, and so are 100 % of the replays of the -m gpu tests).  The
# flag only matters to code that captures hipGraphs; it must be in the environment before the HIP runtime
# initialises, i.e. before the first CUDA call of the process — importing this package first is enough.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
