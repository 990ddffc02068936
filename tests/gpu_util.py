import importlib

PKG = "geometric-multigrid-preconditioners-for-long-range-coulomb-interaction_amd"


def pkg():
    return importlib.import_module(PKG)


def capi():
    return pkg().capi
