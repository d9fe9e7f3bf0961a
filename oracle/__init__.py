"""CPU oracle of the reference datapath -- test infrastructure only."""
