"""The explicit synthetic input tables live in the package (bench.py and smoke() use them too)."""
from tactilesimulation_amd.workloads import push_workload, asset, ASSETS, PUSHER_BLOB  # noqa: F401
