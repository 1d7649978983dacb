from gan_amd.constraints import (aig_constraint_penalties, enforce_aig_constraints,  # noqa: F401
                                 validate_strict_aig)
