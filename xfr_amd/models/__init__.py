"""Host-side mirror of the reference's `xfr.models` package (python/xfr/models/) for the EBP hot path."""
