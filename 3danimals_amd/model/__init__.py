"""Mirror of the reference's ``model`` package for the replaced hot-path modules only:
``model.geometry.{dmtet,skinning,util}`` and ``model.render.{mesh,render,util,light,renderutils}``
(SURVEY.md section 8b).  Predictors, networks, datasets and the Trainer stay the reference's own."""
