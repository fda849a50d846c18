"""CPU oracle -- TEST INFRASTRUCTURE (see oracle/piml_oracle.c).  Never imported by piml_amd/."""
