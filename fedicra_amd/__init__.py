"""fedicra_amd -- MI355X-native drop-in for the FedICRA training hot path."""
__version__ = "0.1.0"
