"""MI355X-native TGN-recommender training path (drop-in for youngandbin/PfoTGNRec's hot path)."""
__version__ = "0.1.0"
