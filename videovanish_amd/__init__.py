"""videovanish_amd: MI355X-native DiffuEraser hot path."""
