"""Drop-in for the reference's ``simple_knn`` package (an absent third-party submodule): ``simple_knn._C.distCUDA2``."""
