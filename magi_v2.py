"""``import magi_v2`` drop-in (the reference's module name, test_magi_script.py:15): re-exports the
MI355X implementation from magi_v2_amd."""
from magi_v2_amd.api import MAGI_v2, logarithmic_temperature_schedule  # noqa: F401
