"""Seeded, NAME-keyed parameter initialisation used by the golden generator and the tests (TEST INFRASTRUCTURE).
The implementation lives in the product package (ctrl-adapter_amd/synthetic.py) so that bench.py's timed leg builds
its synthetic weights without importing anything under oracle/; this module re-exports it for the checker side."""
from ctrl_adapter_amd.synthetic import seeded_init, seeded_tensor  # noqa: F401
