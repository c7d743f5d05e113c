"""similari_amd — MI355X-native association engine for Similari-style trackers (hot path only)."""
__version__ = "0.1.0"
