"""Import-compatibility alias: ``from ku.ebm import RBM, DBN`` resolves to the MI355X build.

Only the ``ku.ebm`` sub-package of the reference is provided (the RBM/DBN hot path); the
reference's other sub-packages are out of scope.
"""
