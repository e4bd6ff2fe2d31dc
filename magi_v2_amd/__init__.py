"""magi_v2_amd -- MI355X-native engine for the MAGI hot path (Matern build, Cholesky, NUTS log-posterior
+ gradient) behind the Python surface of sophiaxxiao/magi_v2.  See DESIGN.md / INTEGRATION.md."""
from .api import MAGI_v2, logarithmic_temperature_schedule  # noqa: F401

__all__ = ["MAGI_v2", "logarithmic_temperature_schedule"]
