"""Stand-in for torch-geometric 1.6.1 (only what the reference touches)."""
