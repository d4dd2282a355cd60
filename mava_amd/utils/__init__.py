"""Host-side utilities behind the reference's module names (mava/utils/): checkpointing, logging."""
