"""CPU oracle of the reference's predicate path -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
