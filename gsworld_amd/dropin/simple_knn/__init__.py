"""Drop-in ``simple_knn`` package (``from simple_knn._C import distCUDA2``)."""
