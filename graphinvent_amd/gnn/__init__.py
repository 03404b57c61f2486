"""Drop-in replacement for the reference's ``graphinvent/gnn`` package (``import gnn.mpnn``).

Put the directory that CONTAINS this package (``graphinvent_amd/``) ahead of the reference's
``graphinvent/`` on ``sys.path`` and ``Workflow.create_model`` (Workflow.py:265-292) picks up
``gnn.mpnn.GGNN`` from here unchanged; or import it as ``graphinvent_amd.gnn.mpnn``."""
