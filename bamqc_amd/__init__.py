"""bamqc_amd — MI355X-native per-read aggregation of BamQC (OverallNumbers + QualityCheck +
TripletCounting) behind a C ABI (include/bamqc.h).  See DESIGN.md."""
from .api import Aggregator, BamQCError, DeviceBatch  # noqa: F401
