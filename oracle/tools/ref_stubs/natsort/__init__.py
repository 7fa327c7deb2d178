"""Import-time stand-in for natsort (osmosis_utils/data.py:7); datasets are not used here."""
natsorted = sorted
