"""marex_amd -- MI355X-native hot path for marEx-style ``preprocess_data``."""
