"""Empty import-time stand-in for cv2 (osmosis_utils/data.py:1 imports it; never used here)."""
