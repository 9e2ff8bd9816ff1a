"""Data harness with the reference's import surface (``dataloaders.fundus_dataloader``,
``dataloaders.custom_transforms``, ``mypath``).  CPU-side, PIL / numpy / scipy only (no cv2).
SURVEY.md ranks the input pipeline as a 'next' row (8f-2): this is the plain harness that lets the
unchanged entry script run, not an optimised loader."""
