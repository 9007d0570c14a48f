"""Stand-in for ogb 1.2.4: only the two feature-dimension constants."""
