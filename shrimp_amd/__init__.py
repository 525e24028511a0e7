"""shrimp_amd -- MI355X-native hot path of SHRiMP2's gmapper (seed lookup + Smith-Waterman extension)."""
