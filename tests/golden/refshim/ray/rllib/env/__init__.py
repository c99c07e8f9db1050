class MultiAgentEnv:
    """Marker base class."""
