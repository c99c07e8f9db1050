"""Stand-in for pygame (rendering is never used by the fixture generator)."""
