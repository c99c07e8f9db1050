"""Stand-in for pygame.freetype."""
