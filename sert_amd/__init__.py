"""sert_amd: MI355X-native execution of SERT's training + entity-scoring hot path."""
