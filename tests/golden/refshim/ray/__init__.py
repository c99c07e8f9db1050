"""Stand-in for ray: only the two names the env modules import."""
