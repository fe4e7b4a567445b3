"""Device-side mirrors of the reference's post-processing operators (reference
bootstrapper/post/)."""
