"""RandLA-Net input pipeline on the MI355X (first piece): the k-NN index pyramid."""
